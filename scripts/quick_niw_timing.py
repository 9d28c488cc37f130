"""Dev helper: time the NIW sweep + statistics kernels on synthetic data (not the bench)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
sorted_pts = (sys.argv[4] != "shuffled") if len(sys.argv) > 4 else True
rng = np.random.default_rng(0)
mus = (rng.normal(size=(K, D)) * 10).astype(np.float32)
z = np.sort(rng.integers(0, K, n)) if sorted_pts else rng.integers(0, K, n)
X = (mus[z] + rng.normal(size=(n, D)).astype(np.float32)).astype(np.float32)
mu3 = np.repeat(mus, 3, axis=0) + rng.normal(size=(3 * K, D)).astype(np.float32) * 0.1
R = np.tile(np.triu(rng.normal(size=(D, D)) * 0.05 + np.eye(D)).astype(np.float32).ravel(), (3 * K, 1))
logdet = np.zeros(3 * K, np.float32)
wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1)
wk.upload_points(X)
wk.set_params_niw_chol(mu3, R, logdet, np.full((K, 2), 0.5, np.float32), np.full(K, 1.0 / K, np.float32))
for it in range(6):
    t0 = time.time(); wk.sweep(it + 1); wk.sync(); t1 = time.time()
    pk = wk.suffstats_packed(); t2 = time.time()
    sm, st = wk.last_kernel_ms()
    flops = 2.0 * n * D * D * (K + 2) + 2.0 * n * D * D
    print(f"it{it}: sweep wall {1e3*(t1-t0):.3f} ms kernel {sm:.3f} ms ({flops/sm/1e9:.1f} TF/s algorithmic) | stats wall {1e3*(t2-t1):.3f} ms kernels {st:.3f} ms")
lab, sub = wk.get_labels()
print("acc", (lab == z + 1).mean(), "bins", np.bincount(lab)[:6])
