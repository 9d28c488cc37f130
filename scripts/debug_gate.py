import sys, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
D, K = 64, 32
N = 200000
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
def report(tag):
    lab, sub = wk.get_labels()
    for k in (0, 7):
        m = lab == k + 1
        Xk = X[m].astype(np.float64)
        p = s.params
        out = []
        for w in (1, 2):
            R = p["R"][3 * k + w].astype(np.float64); mu = p["mu"][3 * k + w].astype(np.float64)
            q = (((Xk - mu) @ R.T) ** 2).sum(1)
            out.append((float(p["logdet"][3 * k + w]), q.mean(), q[sub[m] == 1].mean() if (sub[m]==1).any() else -1, q[sub[m] == 2].mean() if (sub[m]==2).any() else -1))
        print(tag, "cluster", k, "n", m.sum(), "Nl,Nr", (sub[m] == 1).sum(), (sub[m] == 2).sum(), "lr_w", s.lr_weights[k], "\n    l: logdet %.2f q_all %.2f q_on_l %.2f q_on_r %.2f" % out[0], "\n    r: logdet %.2f q_all %.2f q_on_l %.2f q_on_r %.2f" % out[1],
              "\n    post nu", s.post["nu"][3*k:3*k+3], "kappa", s.post["kappa"][3*k:3*k+3], "logdet_psi", s.post["logdet_psi"][3*k:3*k+3])
report("init")
for it in range(2):
    s.sample_clusters()
    s.prior.upload(s.wk, s.params, s.lr_weights, s.weights)
    report(f"it{it} before sweep (params drawn from stats of current labels)")
    s.wk.sweep(s._next_epoch(), False)
    report(f"it{it} after sweep")
    s.update_suff_stats_posterior()
    s.reset_bad_clusters()
