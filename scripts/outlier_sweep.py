"""Dev helper: D = 64 sweep kernel time when a fraction of the points are outliers (uniform in the data's bounding box): every outlier
is a point for which no cluster can be excluded.  python3 scripts/outlier_sweep.py [N] [fractions ...]"""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
fr = [float(v) for v in sys.argv[2:]] or [0.0, 1e-4, 1e-3, 1e-2]
D, K = 64, 32
X0, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
lo, hi = X0.min(0), X0.max(0)
for f in fr:
    X = X0.copy()
    rng = np.random.default_rng(7)
    idx = rng.choice(N, int(f * N), replace=False)
    X[idx] = (lo + (hi - lo) * rng.random((len(idx), D))).astype(np.float32)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
    wk.upload_points(X)
    s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
    s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
    for _ in range(25):
        s.group_step(False, False)
    ms = []
    for _ in range(20):
        s.group_step(False, False); ms.append(wk.last_kernel_ms()[0])
    w = wk.last_sweep_work() if hasattr(wk, "last_sweep_work") else None
    print(f"outliers {f:g}: sweep {np.median(ms):.4f} ms, K = {s.K}", ("work " + str(w)) if w is not None else "", flush=True)
    wk.close()
