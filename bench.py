#!/usr/bin/env python3
"""Headline benchmark: Gibbs iterations/sec of the restricted-Gibbs sweep, NIW prior, D=64,
N=10^7 synthetic points in 32 true components, on N GPUs of one node (BASELINE.json metric).

A "step" is one full `group_step` (host posterior draws -> parameter upload -> fused label +
sub-label sampling kernel -> sort + sufficient statistics kernels -> [all-reduce] -> posterior
update -> split / merge Metropolis steps -> relabel), i.e. exactly the region the reference
times as `iter_count` (src/dp-parallel-sampling.jl:363-366).  One-time work (data generation,
upload, initial labels, burn-in until the split/merge gates are open) is outside the timed
region, as in the reference.

Strong scaling: the N points are fixed and shard over the ranks by contiguous column ranges;
the one data-path collective is the all-reduce of the packed sufficient statistics.

Launch: `python bench.py --gpus 1 ...` or
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W`.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense FP32 matrix peak (no TF32/xf32 on gfx950)
# PMC figures of the D=64 sweep kernel at the bench workload (separate rocprofv3 --pmc passes, medians over launches,
# profiles/r01e_bench_niw_d64_n1e7_pmc.json), per point so that they scale with the shard size:
#   HBM bytes: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE -> (2*1318072.3125 + 81211.0) KiB / 1e7 points
#   executed matrix flops: SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 -> 314696656 x 512 / 1e7 points (K = 32 clusters)
PMC_BYTES_PER_POINT_D64 = (2 * 1318072.3125 + 81211.0) * 1024 / 1e7
PMC_EXECUTED_FLOPS_PER_POINT_D64_K32 = 314696656.0 * 512 / 1e7


def cpu_baseline(host, X_local, D, K, sampler, budget_points):
    """Reference-algorithm CPU restatement (oracle/oracle.py: sweep_numpy_niw) timed on a bounded
    sample of the same workload with the same K parameters; scaled to the full N."""
    from threadpoolctl import threadpool_limits
    from oracle import oracle as orc
    n = min(budget_points, X_local.shape[0])
    idx = np.linspace(0, X_local.shape[0] - 1, n).astype(np.int64)   # spans all components
    Xs = np.ascontiguousarray(X_local[idx])
    p = sampler.params
    inv, _ = host.native.niw_expand(p["R"], want_sigma=False)
    mu = p["mu"].astype(np.float32); invS = inv.reshape(3 * K, -1).astype(np.float32); logdet = p["logdet"].astype(np.float32)
    logw = np.log(sampler.weights); loglr = np.log(sampler.lr_weights)
    u0, u1 = orc.uniforms(1, 1, 0, 0, n)
    cores = max(1, min(host.native._cpu_budget(), 64))     # CPUs this process may really use (cgroup quota), not os.cpu_count()
    with threadpool_limits(limits=cores, user_api="blas"):
        orc.sweep_numpy_niw(Xs[:2000], D, mu, invS, logdet, logw, loglr, u0[:2000], u1[:2000])  # warm
        t0 = time.perf_counter()
        orc.sweep_numpy_niw(Xs, D, mu, invS, logdet, logw, loglr, u0, u1)
        dt = time.perf_counter() - t0
    return dt, n, cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=float, default=1e7, help="total number of points (default: the BASELINE metric's N)")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--clusters", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=800000, help="points of the workload the CPU baseline is timed on (about 10 s of CPU work)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # test hook (not used by the driver): DPMM_BENCH_BACKEND=gloo + DPMM_BENCH_SHARE_DEVICE=1 lets several ranks share
    # one GPU so that the world_size > 1 code path can be exercised on a single-GPU box
    backend = os.environ.get("DPMM_BENCH_BACKEND", "nccl")
    if os.environ.get("DPMM_BENCH_SHARE_DEVICE"):
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"

    from __graft_entry__ import load_package
    pkg = load_package()
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    from dpmmsubclusters_jl_amd.host.comm import default_comm

    N, D, K = int(args.points), args.dim, args.clusters
    comm = default_comm()
    lo, hi = (N * rank) // world, (N * (rank + 1)) // world
    data_seed, sampler_seed, burnout = 12345, 123456789, 20
    X, y = host.gaussian_mixture_shard(N, D, K, 100.0, data_seed, lo, hi)

    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))      # default prior, dp-parallel-sampling.jl:272-274
    wk = pkg.Worker(pkg.PRIOR_NIW, D, hi - lo, first_index=lo, device=local_rank, seed=sampler_seed)
    wk.upload_points(X)
    s = host.DPMMSampler(wk, prior, 10.0, N, sampler_seed, burnout=burnout, comm=comm)
    sub0 = 1 + (np.random.default_rng([data_seed, 7, rank]).integers(0, 2, hi - lo))
    s.start_from_labels(y, sub0, K)
    # burn-in (setup, untimed): `burnout` sweeps until every cluster's split/merge gate is open
    for _ in range(burnout + 1):
        s.group_step(False, False)
    for _ in range(args.warmup):
        s.group_step(False, False)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    sweep_ms, stats_ms, ks = [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s.group_step(False, False)
        a, b = wk.last_kernel_ms()   # HIP events on the library's stream (stream is idle here: stats were read back)
        sweep_ms.append(a); stats_ms.append(b); ks.append(s.K)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=(f"cuda:{local_rank}" if backend == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_local = hi - lo
    k_mean = float(np.mean(ks))
    flops_per_launch = 2.0 * n_local * D * D * (k_mean + 2)       # likelihood vs K clusters + own left/right
    avg_sweep_ms = float(np.mean(sweep_ms))
    achieved = flops_per_launch / (avg_sweep_ms * 1e-3) / 1e12
    out = {
        "metric": "Gibbs iterations/sec, N=10M D=64 NIW" if (N == 10 ** 7 and D == 64) else f"Gibbs iterations/sec, N={N} D={D} NIW",
        "value": args.steps / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"NIW D={D} N={N} synthetic GMM, {K} true components, K_t={k_mean:.1f} live clusters, "
                               f"alpha=10, default NIW prior, steady state after {burnout + 1} burn-in sweeps",
                   "points_per_gpu": n_local, "parallelism": f"points sharded over {world} GPU(s), 1 all-reduce of packed suff-stats per statistics pass"},
        "roofline": {"kernel": "niw_sweep_direct_kernel<4,4,2>" if D == 64 else "niw_sweep_kernel", "bound": "mfma",
                     "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                     "traffic": (PMC_BYTES_PER_POINT_D64 * n_local if D == 64 else None),
                     "avg_launch_ms": avg_sweep_ms, "algorithmic_flops_per_launch": flops_per_launch,
                     # cluster screening skips most of the algorithmic work, which is why `frac` exceeds 1; the flops the
                     # kernel actually EXECUTES on the matrix pipe (PMC, K=32 bench data) over the live launch time:
                     "executed_tflops": (PMC_EXECUTED_FLOPS_PER_POINT_D64_K32 * n_local / (avg_sweep_ms * 1e-3) / 1e12
                                         if (D == 64 and K == 32) else None),
                     "executed_frac": (PMC_EXECUTED_FLOPS_PER_POINT_D64_K32 * n_local / (avg_sweep_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS
                                       if (D == 64 and K == 32) else None),
                     "stats_kernels_ms": float(np.mean(stats_ms))},
        "host_ms_per_step": {k: 1e3 * v / (args.steps + args.warmup + burnout + 1) for k, v in s.timers.items()},
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        dt, n_s, cores = cpu_baseline(host, X, D, K, s, args.cpu_sample)
        out["cpu_baseline"] = {"value": 1.0 / (dt * N / n_s), "unit": "iterations/s", "cores": cores, "kind": "port",
                               "sample": f"worker path (label + sub-label sampling + 3 Float64 statistic passes per cluster) of the "
                                         f"numpy/BLAS restatement on {n_s} of the {N} points, K={K}, {dt:.2f} s measured, scaled linearly to N"}
    if rank == 0:
        print(json.dumps(out))
    wk.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
