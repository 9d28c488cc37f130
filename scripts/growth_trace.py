"""Dev helper: per-iteration times of a growth run from ONE cluster at the headline shape, with K and the sweep's three launch times.
   python3 scripts/growth_trace.py [N] [iters]"""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 140
D, K = 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.init_first_clusters(1)
wk.set_timing(15)
rows = []
for i in range(iters):
    t0 = time.perf_counter(); s.group_step(False, False); dt = time.perf_counter() - t0
    p = wk.last_sweep_parts_ms(); a, b = wk.last_kernel_ms()
    rows.append((i, s.K, 1e3 * dt, a, b, p[0], p[1], p[2]))
for r in rows:
    print("it %3d K %3d  step %.3f ms  sweep %.3f  stats %.3f  (lean %.3f  labels/list %.3f  sub %.3f)" % r)
tot = sum(r[2] for r in rows)
print(f"total {tot:.1f} ms for {iters} iterations = {1e3 * iters / tot:.1f} it/s")
