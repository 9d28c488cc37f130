"""Dev helper: read the phase cycle sums from the DPMM_STAMPS diagnostic build."""
import sys, os, ctypes, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
alt = os.path.abspath("dpmmsubclusters.jl_amd/lib/libdpmmhip_stamps.so")
b.lib_path = lambda: alt
D, n, K = 64, 1000000, 32
rng = np.random.default_rng(0)
mus = (rng.normal(size=(K, D)) * 10).astype(np.float32)
z = np.sort(rng.integers(0, K, n))
X = (mus[z] + rng.normal(size=(n, D)).astype(np.float32)).astype(np.float32)
mu3 = np.repeat(mus, 3, axis=0) + rng.normal(size=(3 * K, D)).astype(np.float32) * 0.1
R = np.tile(np.triu(rng.normal(size=(D, D)) * 0.05 + np.eye(D)).astype(np.float32).ravel(), (3 * K, 1))
wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1)
wk.upload_points(X)
wk.set_params_niw_chol(mu3, R, np.zeros(3 * K, np.float32), np.full((K, 2), 0.5, np.float32), np.full(K, 1.0 / K, np.float32))
for it in range(3):
    wk.sweep(it + 1); wk.sync()
print("kernel ms (stamped build)", wk.last_kernel_ms()[0])
lib = b.load_library()
nw = 4 * 4096
buf = np.zeros((nw, 8), np.uint64)
lib.dpmm_dev_stamps.restype = ctypes.c_int
used = lib.dpmm_dev_stamps(wk._h, buf.ctypes.data_as(ctypes.c_void_p), nw)
d = buf[:used].astype(np.float64)
d = d[d[:, 6] > 0]
names = ["survivors", "refs(full)", "screen", "draw", "phase2", "total"]
print("waves", len(d), "tiles/wave", d[:, 6].mean())
for i, nm in enumerate(names):
    print(f"{nm:10s} cycles/tile {d[:, i].sum() / d[:, 6].sum():10.0f}   share {100 * d[:, i].sum() / d[:, 5].sum():5.1f}%")
print("quad per matrix (32 of them in phase 1):", d[:, 1].sum() / d[:, 6].sum() / 32)
