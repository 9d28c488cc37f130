"""Dev helper: A/B the NIW sweep kernel between library builds on zero-mean data (so a build that
skips the x-mu subtraction still produces sane labels)."""
import sys, os, time, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
alt = os.environ.get("DPMM_LIB")
if alt:
    b.lib_path = lambda: alt
D, n, K = 64, 1000000, 32
rng = np.random.default_rng(0)
scale = (1.3 ** np.arange(K)).astype(np.float32)
z = np.sort(rng.integers(0, K, n))
X = (rng.normal(size=(n, D)).astype(np.float32) * scale[z][:, None]).astype(np.float32)
mu3 = np.zeros((3 * K, D), np.float32)
R = np.stack([np.eye(D, dtype=np.float32) / scale[k // 3] for k in range(3 * K)]).reshape(3 * K, -1)
logdet = np.repeat(2 * D * np.log(scale), 3).astype(np.float32)
wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1)
wk.upload_points(X)
wk.set_params_niw_chol(mu3, R, logdet, np.full((K, 2), 0.5, np.float32), np.full(K, 1.0 / K, np.float32))
ts = []
for it in range(8):
    wk.sweep(it + 1); wk.sync()
    ts.append(wk.last_kernel_ms()[0])
lab, sub = wk.get_labels()
print(os.path.basename(alt or "default"), "kernel ms", " ".join(f"{t:.3f}" for t in ts), "acc", (lab == z + 1).mean())
