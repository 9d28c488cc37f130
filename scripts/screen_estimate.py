"""Dev helper: how many clusters would survive a cheap lower-bound screen per 64-point wave on the bench data?"""
import sys, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N, D, K = 200000, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(5):
    s.group_step(False, False)
p = s.params
R = p["R"].reshape(3 * s.K, D, D).astype(np.float64)[::3]; mu = p["mu"].reshape(3 * s.K, D).astype(np.float64)[::3]
cst = -0.5 * p["logdet"].reshape(-1)[::3].astype(np.float64) + np.log(s.weights.astype(np.float64))
Rinv = np.linalg.inv(R)
lam_frob = 1.0 / (Rinv ** 2).sum((1, 2))                    # >= ... lower bound on lambda_min(Sigma^-1)
lam_true = 1.0 / np.linalg.norm(Rinv, 2, axis=(1, 2)) ** 2
lab, _ = wk.get_labels()
Xd = X.astype(np.float64)
for name, lam in (("frobenius", lam_frob), ("exact", lam_true)):
    survivors = []
    for w0 in range(0, N - 64, 64 * 37):
        xs = Xd[w0:w0 + 64]; k0 = lab[w0] - 1
        z = xs - mu[k0]; q0 = ((z @ R[k0].T) ** 2).sum(1)
        best = cst[k0] - 0.5 * q0                            # reference value per point
        d2 = ((xs[:, None, :] - mu[None, :, :]) ** 2).sum(2)  # (64, K)
        ub = cst[None, :] - 0.45 * lam[None, :] * d2          # upper bound of a_k
        keep = ~np.all(ub < best[:, None] - 50.0, axis=0)
        keep[k0] = True
        survivors.append(keep.sum())
    survivors = np.array(survivors)
    print(name, "clusters evaluated per wave: mean %.2f median %d max %d (of K=%d)" % (survivors.mean(), np.median(survivors), survivors.max(), s.K))
