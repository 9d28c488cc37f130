"""Dev helper: steady-state step time and sweep-kernel time at the C5-like shape (NIW D=256 / D=128, K=32, one GPU shard)."""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 625000
K = 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(5):
    s.group_step(False, False)
ts, sw, st = [], [], []
for _ in range(10):
    t0 = time.perf_counter(); s.group_step(False, False); ts.append(time.perf_counter() - t0)
    a, b = wk.last_kernel_ms(); sw.append(a); st.append(b)
lab, _ = wk.get_labels()
print(f"D={D} N={N} K={s.K}: step {1e3 * np.mean(ts):.2f} ms, sweep kernel {np.mean(sw):.2f} ms, stats kernels {np.mean(st):.2f} ms, "
      f"label agreement with generator {np.mean(lab == y):.4f}")
steps = 15 + 1
print({k: round(1e3 * v / steps, 2) for k, v in s.timers.items()})
