import sys, time, importlib, gc
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N, D, K = 1250000, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for mode in ("gc on", "gc off"):
    if mode == "gc off":
        gc.collect(); gc.disable()
    ts = []
    for _ in range(60):
        t0 = time.perf_counter(); s.group_step(False, False); ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts)
    print(mode, "median %.2f  mean %.2f  max %.2f  n>10ms %d" % (np.median(ts), ts.mean(), ts.max(), (ts > 10).sum()), "slow at", np.flatnonzero(ts > 10).tolist())
