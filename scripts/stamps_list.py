"""Dev helper: phase stamps (DPMM_STAMPS build) of the LIST instantiation of niw_sweep_direct_kernel behind niw_lean_kernel on the bench data:
which phase a slow handed-on tile spends its time in.  Needs docs/experiments/r05_list_kernel_stamps.patch applied before
scripts/build_stamps.sh (the label-storing instantiations leave the tile loop in front of the stamps' accumulation).
   python3 scripts/stamps_list.py [N]"""
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
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(30):
    s.group_step(False, False)
lib = b.load_library()
lib.dpmm_dev_stamps.restype = ctypes.c_int
nw = 4 * 4096
names = ["x load", "refs(full)", "screen setup", "K-loop", "survivors", "draw", "phase2", "total"]
wk.set_timing(15)
for it in range(24):
    s.group_step(False, False)
    p = wk.last_sweep_parts_ms(); w = wk.last_sweep_work()
    buf = np.zeros((nw, 16), np.uint64)
    used = lib.dpmm_dev_stamps(wk._h, buf.ctypes.data_as(ctypes.c_void_p), nw)
    d = buf[:2048].astype(np.float64); d = d[d[:, 8] > 0]
    if len(d) == 0:
        print(f"sweep {it}: labels launch {p[1]:.3f} ms, no tile handed on"); continue
    i = int(np.argmax(d[:, 7]))
    print(f"sweep {it}: labels launch {p[1]:.3f} ms, {len(d)} waves with a tile; slowest wave: " + ", ".join(f"{nm} {int(d[i, j])}" for j, nm in enumerate(names)) +
          f" | tiles {int(d[i, 8])} tail-screened {int(d[i, 9])} mfma-screened {int(d[i, 10])}; top screens {w['bf16_top_screens']:.0f} screens16 {w['screens16']:.0f} full {w['full_evals']:.0f}")
