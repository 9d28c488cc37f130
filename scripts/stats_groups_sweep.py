"""Dev helper: statistics-pass time against DPMM_OPT_STATS_GROUPS.  python3 scripts/stats_groups_sweep.py D N groups..."""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
from dpmmsubclusters_jl_amd import binding
D, N, K = int(sys.argv[1]), int(float(sys.argv[2])), 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(8):
    s.group_step(False, False)
for g in [int(v) for v in sys.argv[3:]]:
    wk.set_option(binding.OPT_STATS_GROUPS, g)
    ms = []
    for _ in range(6):
        s.group_step(False, False); ms.append(wk.last_kernel_ms()[1])
    print(f"groups {g:5d}: statistics kernels median {np.median(ms[1:]):.3f} ms  min {np.min(ms[1:]):.3f}", flush=True)
