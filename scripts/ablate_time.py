"""Dev helper: sweep / statistics kernel times of an experimental library build (scripts/build_variant.sh) on the bench shape.
   python3 scripts/variant_time.py <lib name or path> [N] [MixtureVar = 100]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
name = sys.argv[1]
alt = name if os.path.exists(name) else os.path.abspath(f"dpmmsubclusters.jl_amd/lib/libdpmmhip_{name}.so")
if name != "default":
    b.lib_path = lambda: alt
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 7
D, K = 64, 32
mixvar = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
X, y = host.gaussian_mixture_shard(N, D, K, mixvar, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(30 if mixvar >= 50 else 70):          # (overlapping clusters: the regime switches settle within ~50 sweeps)
    s.group_step(True, False)
sw, st = [], []
for _ in range(20):
    s.group_step(True, False); a, c = wk.last_kernel_ms(); sw.append(a); st.append(c)
print(f"{name}: sweep kernel median {np.median(sw):.4f} ms  min {np.min(sw):.4f}; statistics pass median {np.median(st):.4f} ms")
