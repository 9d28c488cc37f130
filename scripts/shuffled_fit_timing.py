"""Dev helper: steady-state sweep time on SHUFFLED data (labels mixed inside every wave without re-ordering)."""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N, D, K = 1000000, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
perm = np.random.default_rng(1).permutation(N)
X = np.ascontiguousarray(X[perm]); y = y[perm]
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
ts, ks = [], []
for it in range(12):
    t0 = time.perf_counter(); s.group_step(False, False); ts.append(1e3 * (time.perf_counter() - t0)); ks.append(wk.last_kernel_ms()[0])
print("step ms", " ".join(f"{t:.2f}" for t in ts))
print("sweep kernel ms", " ".join(f"{t:.2f}" for t in ks))
lab, _ = wk.get_labels()
print("acc", (lab == y).mean(), "K", s.K)
