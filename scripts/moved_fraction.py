import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N, D, K = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(60):
    s.group_step(False, False)
l0, s0 = wk.get_labels()
for it in range(6):
    s.group_step(False, False)
    l1, s1 = wk.get_labels()
    print(f"step {it}: labels changed {np.mean(l1 != l0):.5f}  sub-labels changed (same label) {np.mean((s1 != s0) & (l1 == l0)):.4f}  bins changed {np.mean((l1 != l0) | (s1 != s0)):.4f}  bad resets so far {s.model.get('counters')[4:6]}")
    l0, s0 = l1, s1
Nk = s.N
mn = np.minimum(Nk[:, 1], Nk[:, 2])
print("sum over clusters of min(N_l, N_r) / N =", float(mn.sum() / Nk[:, 0].sum()), "; per cluster min share:", np.round(np.sort(mn / np.maximum(Nk[:, 0], 1)), 3).tolist())
