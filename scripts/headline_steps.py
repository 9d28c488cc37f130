"""Dev helper: per-step wall times of the bench's headline block (same set-up, same timing mode and per-step event reads).
   python3 scripts/headline_steps.py [steps=30] [warmup=5] [timing_in_warmup=0|1]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from __graft_entry__ import load_package
import bench
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N, D, K = 10 ** 7, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, bench.DATA_SEED, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, first_index=0, device=0, seed=bench.SAMPLER_SEED)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, bench.ALPHA, N, bench.SAMPLER_SEED, burnout=bench.BURNOUT)
s.start_from_labels(y, 1 + (np.random.default_rng([bench.DATA_SEED, 7, 0]).integers(0, 2, N)), K)
for _ in range(bench.BURNOUT + 1 + 100):
    s.group_step(False, False)
if tw:
    wk.set_timing(1)
for _ in range(warm):
    s.group_step(False, False)
    if tw:
        wk.last_kernel_ms()
torch.cuda.synchronize(); wk.sync()
wk.set_timing(1)
wk.last_sweep_work()
ts, km = [], []
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    s.group_step(False, False)
    km.append(wk.last_kernel_ms()[0])
    ts.append(1e3 * (time.perf_counter() - a))
torch.cuda.synchronize(); wk.sync()
el = time.perf_counter() - t0
print(f"headline block: {steps / el:.1f} it/s ({1e3 * el / steps:.4f} ms per step); sweep kernel median {np.median(km):.4f} ms; per step ms: " + " ".join(f"{t:.2f}" for t in ts))
ts2 = []
for _ in range(steps):
    a = time.perf_counter(); s.group_step(False, False); ts2.append(1e3 * (time.perf_counter() - a))
wk.sync()
print("next block without event reads: median %.3f max %.3f" % (np.median(ts2), np.max(ts2)))
