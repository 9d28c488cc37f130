"""Dev helper: phase cycles of niw_post_lds_kernel (needs a -DDPMM_POST_STAMPS build of the worker library: python3 scripts/post_stamps.py D lib.so)."""
import sys, numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
import importlib, os
if len(sys.argv) > 2:
    b = importlib.import_module("dpmmsubclusters_jl_amd.binding"); alt = os.path.abspath(sys.argv[2]); b.lib_path = lambda: alt
D, n, K = int(sys.argv[1]), 200000, 32
rng = np.random.default_rng(0)
X = (rng.normal(size=(n, D)) + rng.normal(size=(K, D))[rng.integers(0, K, n)] * 4).astype(np.float32)
wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1)
wk.upload_points(X); wk.set_labels(rng.integers(1, K + 1, n), rng.integers(1, 3, n)); wk.set_num_clusters(K)
wk.master_setup(1.0, D + 3.0, np.zeros(D), np.eye(D)); wk.suffstats_device(None)
slots = np.arange(K, dtype=np.int32)
for _ in range(3): got = wk.master_posterior(None, slots)
g = got.reshape(-1, 4)
if D <= 128:
    print("form %d diag %d panel %d trailing %d | epilogue %d total %d (shader cycles)" % (g[5,0], g[5,1], g[5,2], g[5,3], g[6,0], g[6,1]))
else:
    print("factorisation kernel: diag %d panel %d trailing %d total %d (shader cycles)" % (g[5,0], g[5,1], g[5,2], g[5,3]))

lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
for ep in range(3): wk.master_draw(ep + 1, slots, lr, w)
mu, R, ld = wk.master_draws(K)
print("draw: noise %d  stage L %d  products %d  triangular %d | logdet + mean %d  total %d (shader cycles)" % tuple(ld.reshape(-1)[3:9]))
