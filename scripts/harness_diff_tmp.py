import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
if len(sys.argv) > 1 and sys.argv[1] == 'avail':
    print('torch.cuda.is_available()', torch.cuda.is_available())
from __graft_entry__ import load_package
import bench
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
from dpmmsubclusters_jl_amd import binding
N, D, K = 10 ** 7, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, bench.DATA_SEED, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
sub0 = 1 + (np.random.default_rng([bench.DATA_SEED, 7, 0]).integers(0, 2, N))
def run(tag, timeout_opt, use_comm):
    wk = pkg.Worker(pkg.PRIOR_NIW, D, N, first_index=0, device=0, seed=bench.SAMPLER_SEED)
    wk.upload_points(X)
    if timeout_opt:
        wk.set_option(binding.OPT_COMM_TIMEOUT_MS, 1e3 * 300.0)
    kw = dict(comm=host.LocalComm()) if use_comm else {}
    s = host.DPMMSampler(wk, prior, bench.ALPHA, N, bench.SAMPLER_SEED, burnout=bench.BURNOUT, **kw)
    s.start_from_labels(y, sub0, K)
    for _ in range(bench.BURNOUT + 1 + 100 + 5):
        s.group_step(False, False)
    torch.cuda.synchronize(); wk.sync()
    wk.set_timing(1); wk.last_sweep_work()
    km = []
    t0 = time.perf_counter()
    for _ in range(30):
        s.group_step(False, False); km.append(wk.last_kernel_ms()[0])
    torch.cuda.synchronize(); wk.sync()
    el = time.perf_counter() - t0
    print(f"{tag}: {30 / el:.1f} it/s, sweep kernel median {np.median(km):.4f} ms", flush=True)
    wk.close()
run("run", False, False)
