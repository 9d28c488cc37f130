"""Dev helper: the native master's per-sweep maths in isolation (no GPU): posterior + factorisation of the 3K distributions
(dpmmh_model_set "packed" -> ingest), parameter draws + gates (dpmmh_sample_clusters) and the merge-pair marginals.
   python3 scripts/host_micro.py [D] [K] [threads]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from __graft_entry__ import load_package
load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
from fake_worker import FakeWorker
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nt = int(sys.argv[3]) if len(sys.argv) > 3 else host.native.default_threads()
rng = np.random.default_rng(0)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3.0, np.eye(D))
wk = FakeWorker(0, D, 0)
s = host.DPMMSampler(wk, prior, 10.0, 10 ** 7, seed=7, burnout=20, nthreads=nt)
s._configure()
stride = 1 + D + D * (D + 1) // 2
packed = np.zeros((2 * K, stride)); il = np.tril_indices(D)
for r in range(2 * K):
    n = 150000
    A = rng.normal(size=(D + 8, D)); C = A.T @ A / (D + 8) + 0.1 * np.eye(D); c = rng.normal(size=D) * 10
    packed[r, 0] = n; packed[r, 1:1 + D] = n * c; packed[r, 1 + D:] = (n * (C + np.outer(c, c)))[il]
s.model.set("K", K)
def timeit(fn, reps=30):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * np.median(ts), 1e3 * np.min(ts)
print(f"D={D} K={K} threads={nt}")
print("posterior (ingest) ms  median %.3f  min %.3f" % timeit(lambda: s.model.set("packed", packed)))
print("sample_clusters   ms  median %.3f  min %.3f" % timeit(lambda: s.model.sample_clusters()))
s.model.set("splittable", np.ones(K, np.uint8))
print("merge ratios      ms  median %.3f  min %.3f" % timeit(lambda: s.model.debug_merge_log_hr(), 10))
def back_to_back():
    s.model.set("packed", packed); s.model.sample_clusters()
print("posterior+sample  ms  median %.3f  min %.3f" % timeit(back_to_back))
def after_sleep():
    time.sleep(0.003); t0 = time.perf_counter(); s.model.set("packed", packed); return time.perf_counter() - t0
after_sleep(); print("posterior after a 3 ms sleep (pool asleep) ms median %.3f" % (1e3 * np.median([after_sleep() for _ in range(30)])))
