"""Dev helper: time the device master's calls (posterior + factorisation, draw + hand-over) for K clusters.  python3 scripts/master_timing.py D n K"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
if len(sys.argv) > 4:      # alternative build of the worker library (diagnostic variants)
    import os
    b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
    alt = os.path.abspath(sys.argv[4]); b.lib_path = lambda: alt
D, n, K = int(sys.argv[1]), int(float(sys.argv[2])), int(sys.argv[3])
rng = np.random.default_rng(0)
X = (rng.normal(size=(n, D)) + rng.normal(size=(K, D))[rng.integers(0, K, n)] * 4).astype(np.float32)
wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1)
wk.upload_points(X)
wk.set_labels(rng.integers(1, K + 1, n), rng.integers(1, 3, n))
wk.set_num_clusters(K)
wk.master_setup(1.0, D + 3.0, np.zeros(D), np.eye(D))
wk.suffstats_device(None)
slots = np.arange(K, dtype=np.int32)
lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
def best(f, reps=10):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); wk.sync(); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts), 1e3 * float(np.median(ts))
print("D=%d K=%d  posterior (form + factorise, blocking): min %.3f ms median %.3f ms" % ((D, K) + best(lambda: wk.master_posterior(None, slots))))
ep = [0]
def draw():
    ep[0] += 1; wk.master_draw(ep[0], slots, lr, w)
print("D=%d K=%d  draw + hand-over (to stream idle):       min %.3f ms median %.3f ms" % ((D, K) + best(draw)))
