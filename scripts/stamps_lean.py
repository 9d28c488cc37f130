"""Dev helper: phase stamps of niw_lean_kernel (DPMM_STAMPS build, scripts/build_stamps.sh) on the bench data.
   python3 scripts/stamps_lean.py [N]"""
import sys, os, ctypes, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
alt = os.path.abspath("dpmmsubclusters.jl_amd/lib/libdpmmhip_stamps.so")
b.lib_path = lambda: alt
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
D, K = 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(30):
    s.group_step(False, False)
print("sweep ms (stamped build)", wk.last_kernel_ms()[0])
lib = b.load_library()
nw = 4 * 4096
buf = np.zeros((nw, 16), np.uint64)
lib.dpmm_dev_stamps.restype = ctypes.c_int
lib.dpmm_dev_stamps(wk._h, buf.ctypes.data_as(ctypes.c_void_p), nw)
d = buf[8192:8192 + 2048].astype(np.float64); d = d[d[:, 8] > 0]
nt = d[:, 8].sum()
for i, nm in enumerate(["x gather (to vmcnt 0)", "tail feats + conversion", "bracket", "ball / tail / bottom screens", "head request + uniforms", "sub-label evaluation + store"]):
    print(f"{nm:32s} cycles/tile {d[:, i].sum() / nt:9.0f}   share {100 * d[:, i].sum() / d[:, 7].sum():5.1f}%")
print(f"total cycles/tile {d[:, 7].sum() / nt:.0f}; tiles per wave {d[:, 8].min():.0f}..{d[:, 8].max():.0f}")
