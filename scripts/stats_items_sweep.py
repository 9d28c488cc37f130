"""Dev helper: statistics-pass time against DPMM_OPT_STATS_ITEMS x DPMM_OPT_STATS_GROUPS (the pass includes the sort, the reduce and the derivation).
   python3 scripts/stats_items_sweep.py D N items,items,... groups,groups,..."""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
from dpmmsubclusters_jl_amd import binding
D, N, K = int(sys.argv[1]), int(float(sys.argv[2])), 32
items = [int(v) for v in sys.argv[3].split(",")]
groups = [int(v) for v in sys.argv[4].split(",")]
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
for it in items:
    wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
    wk.set_option(binding.OPT_STATS_ITEMS, it)            # (before the first parameters: it sizes the slabs)
    wk.upload_points(X)
    s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
    s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
    for _ in range(25):
        s.group_step(False, False)
    for g in groups:
        wk.set_option(binding.OPT_STATS_GROUPS, g)
        ms = []
        for _ in range(7):
            s.group_step(False, False); ms.append(wk.last_kernel_ms()[1])
        print(f"items {it:6d} groups {g:5d}: statistics pass median {np.median(ms[2:]):.4f} ms  min {np.min(ms[2:]):.4f}", flush=True)
    wk.close()
