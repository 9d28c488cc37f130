"""Dev helper: sweep-kernel time of an experimental library build against the number of workgroups (no splits: the build may draw wrong sub-labels).
   python3 scripts/ablate_grid.py <lib name> N grids..."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
name = sys.argv[1]
if name != "default":
    alt = os.path.abspath(f"dpmmsubclusters.jl_amd/lib/libdpmmhip_{name}.so")
    b.lib_path = lambda: alt
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N, D, K = int(float(sys.argv[2])), 64, 32
M = int(os.environ.get("PERIODIC", "0"))        # M > 0: X[i] = X[i % M] (a gather of row i % M then reads the same values from cache)
if M:
    Xb, yb = host.gaussian_mixture_shard(M, D, K, 100.0, 12345, 0, M)
    reps = (N + M - 1) // M
    X = np.tile(Xb, (reps, 1))[:N].copy(); y = np.tile(yb, reps)[:N].copy()
else:
    X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(12):
    s.group_step(True, False)
for g in [int(v) for v in sys.argv[3:]]:
    wk.set_option(b.OPT_SWEEP_GRID, g)
    ms = []
    for _ in range(8):
        s.group_step(True, False); ms.append(wk.last_kernel_ms()[0])
    print(f"{name} grid {g:5d}: sweep kernel median {np.median(ms[1:]):.3f} ms  min {np.min(ms[1:]):.3f}", flush=True)
