"""Dev helper: how many clusters have an empty sub-cluster per step (reset_bad_clusters traffic)?"""
import sys, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
D, K = 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
orig = s.reset_bad_clusters
def spy():
    bad = np.flatnonzero((s.N[:, 1] == 0) | (s.N[:, 2] == 0))
    small = np.minimum(s.N[:, 1], s.N[:, 2])
    print(f"bad={len(bad)} K={s.K} min(Nl,Nr) quantiles: {np.sort(small)[:8].astype(int).tolist()} N of bad: {s.N[bad,0].astype(int).tolist()[:6]}")
    orig()
s.reset_bad_clusters = spy
for it in range(40):
    s.group_step(False, False)
