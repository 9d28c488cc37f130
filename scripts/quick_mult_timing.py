"""Dev helper: time the Multinomial sweep + statistics kernels (C4: D=1000, N=1e6, K=32)."""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(0)
P = rng.dirichlet(np.ones(D) * 0.5, size=K)
z = rng.integers(0, K, n)
X = np.empty((n, D), np.float32)
for k in range(K):
    m = z == k
    X[m] = rng.multinomial(100, P[k], size=int(m.sum()))
logp = np.log(np.maximum(np.repeat(P, 3, axis=0), 1e-30)).astype(np.float32)
wk = pkg.Worker(pkg.PRIOR_MULT, D, n, device=0, seed=1)
wk.upload_points(X)
wk.set_params_mult(logp, np.full((K, 2), 0.5, np.float32), np.full(K, 1.0 / K, np.float32))
for it in range(5):
    t0 = time.time(); wk.sweep(it + 1); wk.sync(); t1 = time.time()
    pk = wk.suffstats_packed(); t2 = time.time()
    sm, st = wk.last_kernel_ms()
    byts = 4.0 * n * D + 16.0 * n
    print(f"it{it}: sweep kernel {sm:.3f} ms ({byts/sm/1e6:.0f} GB/s algorithmic) | stats kernels {st:.3f} ms wall {1e3*(t2-t1):.3f}")
lab, _ = wk.get_labels()
print("acc", (lab == z + 1).mean())
