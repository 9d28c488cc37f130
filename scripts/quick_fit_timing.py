"""Dev helper: steady-state iteration timing breakdown (host vs GPU) -- not the bench."""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
t0 = time.time()
x, y, _, _ = host.generate_gaussian_data(N, D, K, 100.0, seed=12345)
print("gen", time.time() - t0)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(np.ascontiguousarray(x.T))
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
rng = np.random.default_rng(0)
s.start_from_labels(y.astype(np.int64), rng.integers(1, 3, N), K)
ic, _, _, kh = s.run_model(iters, 1, verbose=False)
print("iter ms:", " ".join(f"{1e3*t:.2f}" for t in ic))
print("K:", kh)
tot = sum(ic)
print("timers (ms/iter):", {k: round(1e3 * v / iters, 3) for k, v in s.timers.items()}, "sum", round(1e3 * sum(s.timers.values()) / iters, 3), "total", round(1e3 * tot / iters, 3))
print("it/s steady:", 1.0 / np.mean(ic[5:-6]))
