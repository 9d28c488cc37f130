"""Dev helper: same kernel, same instruction stream, zero vs random data -> is the gap DVFS?"""
import sys, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
D, n, K = 64, 1000000, 32
rng = np.random.default_rng(0)
for mode in ("random", "zeros", "random"):
    X = (rng.normal(size=(n, D)).astype(np.float32) * 3) if mode == "random" else np.zeros((n, D), np.float32)
    mu3 = (rng.normal(size=(3 * K, D)).astype(np.float32)) if mode == "random" else np.zeros((3 * K, D), np.float32)
    Rm = np.triu(rng.normal(size=(3 * K, D, D)).astype(np.float32) * 0.1 + np.eye(D, dtype=np.float32)) if mode == "random" else np.zeros((3 * K, D, D), np.float32)
    wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1)
    wk.upload_points(X)
    # identical weights everywhere; labels_only-like workload is not needed: zero data gives one label (argmax first) per wave
    wk.set_params_niw_chol(mu3, Rm.reshape(3 * K, -1), np.zeros(3 * K, np.float32), np.full((K, 2), 0.5, np.float32), np.full(K, 1.0 / K, np.float32))
    ts = []
    for it in range(8):
        wk.sweep(it + 1, final=True); wk.sync(); ts.append(wk.last_kernel_ms()[0])
    lab, _ = wk.get_labels()
    print(mode, "kernel ms", " ".join(f"{t:.3f}" for t in ts), "distinct labels", len(np.unique(lab)))
    wk.close()
