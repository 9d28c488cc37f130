"""Dev helper: phase stamps (DPMM_STAMPS build) on the real bench data/parameters."""
import sys, os, ctypes, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
alt = os.path.abspath("dpmmsubclusters.jl_amd/lib/libdpmmhip_stamps.so")
b.lib_path = lambda: alt
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N, D, K = 1000000, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(6):
    s.group_step(False, False)
print("kernel ms (stamped build)", wk.last_kernel_ms()[0])
lib = b.load_library()
nw = 4 * 4096
buf = np.zeros((nw, 16), np.uint64)
lib.dpmm_dev_stamps.restype = ctypes.c_int
used = lib.dpmm_dev_stamps(wk._h, buf.ctypes.data_as(ctypes.c_void_p), nw)
d = buf[:used].astype(np.float64); d = d[d[:, 8] > 0]
names = ["x load", "refs(full)", "screen setup", "K-loop", "survivors", "draw", "phase2", "total"]
nt = d[:, 8].sum()
for i, nm in enumerate(names):
    print(f"{nm:14s} cycles/tile {d[:, i].sum() / nt:10.0f}   share {100 * d[:, i].sum() / d[:, 7].sum():5.1f}%")
print("refs setup (table init, prev labels, first fragments, pre-screen norms) cycles/tile:", d[:, 12].sum() / nt)
print("  of which: table init + k0:", d[:, 13].sum() / nt, " first fragments + pre-screen constants arrive:", d[:, 14].sum() / nt)
print("tail-screened clusters per tile:", d[:, 9].sum() / nt)
print("MFMA-screened clusters per tile:", d[:, 10].sum() / nt)
