"""Dev helper: Multinomial sweep kernel time of an experimental library build (scripts/build_variant.sh) at the C4 shape, on FIXED labels and
parameters (restored before every sweep, so that a variant that computes garbage is still timed on the steady-state row-block mix).
   python3 scripts/mult_variant_time.py <lib name | default> [N] [D] [K]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, ".")
import torch
from __graft_entry__ import load_package
import bench
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
name = sys.argv[1]
alt = name if os.path.exists(name) else os.path.abspath(f"dpmmsubclusters.jl_amd/lib/libdpmmhip_{name}.so")
if name != "default":
    b.lib_path = lambda: alt
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 6
D = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 32
X, y = bench.gpu_multinomial_mixture(torch, N, D, K, 100, 12345)
sub = 1 + np.random.default_rng(0).integers(0, 2, N)
wk = pkg.Worker(pkg.PRIOR_MULT, D, N, device=0, seed=123456789)
torch.cuda.synchronize()
wk.upload_points_device(X.data_ptr(), X.stride(0))
wk.set_labels(y, sub); wk.set_num_clusters(K)
pk = wk.suffstats_packed()
l, r = pk[0::2, 1:], pk[1::2, 1:]
rows = np.stack([l + r, l, r], axis=1).reshape(3 * K, D) + 1.0
logp = np.log(rows / rows.sum(1, keepdims=True)).astype(np.float32)
lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
sw = []
for it in range(8):
    wk.set_labels(y, sub)
    wk.suffstats_packed()
    wk.set_params_mult(logp, lr, w)
    wk.sweep(it + 1)
    sw.append(wk.last_kernel_ms()[0])
lab, _ = wk.get_labels()
print(f"{name}: sweep kernel median {np.median(sw[2:]):.4f} ms  min {np.min(sw):.4f}; labels equal to the start {np.mean(lab == y):.4f}")
