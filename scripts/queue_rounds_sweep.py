"""Dev helper: D <= 64 sweep kernel time against DPMM_OPT_SWEEP_QUEUE_ROUNDS.  python3 scripts/queue_rounds_sweep.py N v1 v2 ..."""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N = int(float(sys.argv[1])); vals = [float(v) for v in sys.argv[2:]]
D, K = 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(25):
    s.group_step(False, False)
res = {v: [] for v in vals}
opt = 13
for step in range(40 * len(vals)):          # the configuration changes EVERY step: drift of the chain (which tiles are expensive) hits all alike
    v = vals[step % len(vals)]
    wk.set_option(opt, v)
    s.group_step(False, False)
    res[v].append(wk.last_kernel_ms()[0])
print(f"N={N}: sweep kernel ms by queue rounds (median / mean of 40 interleaved steps):", {v: (round(float(np.median(x)), 4), round(float(np.mean(x)), 4)) for v, x in res.items()})
