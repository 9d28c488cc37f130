#!/usr/bin/env python3
"""Headline benchmark: Gibbs iterations/sec of the restricted-Gibbs sweep, NIW prior, D=64,
N=10^7 synthetic points in 32 true components, on N GPUs of one node (BASELINE.json metric).

A "step" is one full `group_step` (native master: posterior draws -> parameter hand-over -> fused label +
sub-label sampling kernel -> sort + sufficient statistics kernels -> [RCCL all-reduce] -> posterior
update -> split / merge Metropolis steps -> relabel), i.e. exactly the region the reference
times as `iter_count` (src/dp-parallel-sampling.jl:363-366).  One-time work (data generation,
upload, initial labels, burn-in until the split/merge gates are open) is outside the timed
region, as in the reference.

Strong scaling: the N points are fixed and shard over the ranks by contiguous column ranges;
the one data-path collective is the all-reduce of the packed sufficient statistics (inside libdpmmhip.so).

Besides the contract fields the JSON line carries
  roofline      the dominant kernel (NIW sweep) against the FP32-MFMA peak: `achieved` = ALGORITHMIC flops / live launch time
                (exceeds the peak because exact cluster screening skips work), `frac` = EXECUTED flops / time / peak with the
                executed work counted ON THE DEVICE in the timed launches (dpmm_last_sweep_work), `dense_*` = the same
                kernel with screening switched off (every cluster evaluated in full) in the same process;
  blocks        min / median / max it/s over repeated blocks of `--steps` steps (the headline `value` is the first block);
  growth        a run of the same data from ONE initial cluster (`init_clusters=1`): whole-run and last-20 it/s + K history;
  cpu_baseline  the reference algorithm's worker path on the host cores, P worker processes (see oracle/cpu_baseline.py).

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=float, default=1e7, help="total number of points (default: the BASELINE metric's N)")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--clusters", type=int, default=32)
    ap.add_argument("--blocks", type=int, default=5, help="extra timed blocks of --steps steps after the headline block (min/median/max)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-growth", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--growth-iters", type=int, default=100)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-clock budget of the CPU baseline sample")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"

    from __graft_entry__ import load_package
    pkg = load_package()
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    from dpmmsubclusters_jl_amd.host.comm import default_comm
    from dpmmsubclusters_jl_amd import binding

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
    # ... and `settle` more (setup, untimed): the first ~100 steps of a process run ~2 % slower than all later ones (clock / power state of a
    # GPU that was idle during the upload; measured with --warmup 5 against --warmup 100 on one box: 338.9 against 344.4 it/s over the same
    # 30 timed steps, 200-step blocks afterwards 345-347 either way) -- the metric is the steady-state rate of a long run
    settle = 100
    for _ in range(settle):
        s.group_step(False, False)
    for _ in range(args.warmup):
        s.group_step(False, False)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_block(nsteps, collect=None):
        fence()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            s.group_step(False, False)
            if collect is not None:
                collect()
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    n_local = hi - lo
    sweep_ms, stats_ms, ks, work = [], [], [], []

    def collect():     # HIP events on the library's stream (the stream is idle here: stats were read back)
        a, b = wk.last_kernel_ms()
        sweep_ms.append(a); stats_ms.append(b); ks.append(s.K)

    t_before = dict(s.timers)
    wk.last_sweep_work()              # clear the device's work counters: they add up over the timed launches and are read once afterwards
    elapsed = timed_block(args.steps, collect)
    work.append(wk.last_sweep_work())     # per-launch averages over exactly the timed launches
    t_after = dict(s.timers)
    block_rates = []
    for _ in range(max(0, args.blocks)):
        block_rates.append(args.steps / timed_block(args.steps))

    k_mean = float(np.mean(ks))
    flops_alg = 2.0 * n_local * D * D * (k_mean + 2)       # likelihood vs K clusters + own left/right (per point x points)
    avg_sweep_ms = float(np.mean(sweep_ms))
    exe = float(np.mean([w["executed_flops"] for w in work]))
    achieved = flops_alg / (avg_sweep_ms * 1e-3) / 1e12
    # HBM bytes per launch of the sweep kernel: PMC counters need rocprofv3, so the figure comes from the committed counter summary of
    # THIS command on this round's final build (profiles/, collected by scripts/collect_profiles.sh: FETCH_SIZE doubled per the gfx950
    # note of MI355X_MICROARCH.md + WRITE_SIZE, in KiB) -- only for the configuration it was collected on, else null
    traffic, traffic_source = None, None
    pmc_file = os.path.join(ROOT, "profiles", "r02f_bench_pmc_summary.json")
    if N == 10 ** 7 and D == 64 and world == 1 and os.path.exists(pmc_file):
        try:
            pm = json.load(open(pmc_file))
            for name, c in pm.items():
                if "niw_sweep_direct_kernel" in name:
                    traffic = (2.0 * c["FETCH_SIZE"]["median"] + c["WRITE_SIZE"]["median"]) * 1024.0
                    traffic_source = "profiles/r02f_bench_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, median per launch)"
        except (OSError, ValueError, KeyError):
            traffic = None
    roof = {"kernel": "niw_sweep_direct_kernel<4,4,2,true>" if D <= 64 else "niw_sweep_kernel", "bound": "mfma",
            "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": exe / (avg_sweep_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": 4.0 * n_local * D + 4.0 * n_local,
            "avg_launch_ms": avg_sweep_ms, "algorithmic_flops_per_launch": flops_alg, "executed_flops_per_launch": exe,
            "executed_tflops": exe / (avg_sweep_ms * 1e-3) / 1e12, "pruning_factor": flops_alg / exe if exe else None,
            "algorithmic_frac": achieved / PEAK_F32_MFMA_TFLOPS,
            "work_per_launch": {k: float(np.mean([w[k] for w in work])) for k in ("wave_tiles", "full_evals", "screens16", "tail_pairs")},
            "stats_kernels_ms": float(np.mean(stats_ms))}

    # same kernel, same process, screening off: every cluster is evaluated in full (labels are bit-identical by construction)
    if not args.no_dense:
        wk.set_option(binding.OPT_SCREEN_MARGIN, 0.0)
        dm, dw = [], []
        for _ in range(3):
            s.group_step(False, False)
            dm.append(wk.last_kernel_ms()[0]); dw.append(wk.last_sweep_work()["executed_flops"])
        wk.set_option(binding.OPT_SCREEN_MARGIN, 50.0)
        d_ms, d_fl = float(np.mean(dm[1:])), float(np.mean(dw[1:]))
        roof.update({"dense_launch_ms": d_ms, "dense_executed_tflops": d_fl / (d_ms * 1e-3) / 1e12,
                     "dense_frac": d_fl / (d_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "dense_algorithmic_tflops": flops_alg / (d_ms * 1e-3) / 1e12})

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
                               f"alpha=10, default NIW prior, steady state after {burnout + 1} burn-in + {settle} settling sweeps",
                   "points_per_gpu": n_local, "parallelism": f"points sharded over {world} GPU(s), 1 RCCL all-reduce of packed suff-stats per statistics pass"},
        "roofline": roof,
        "blocks": {"it_per_s": block_rates, "min": float(np.min(block_rates)) if block_rates else None,
                   "median": float(np.median(block_rates)) if block_rates else None, "max": float(np.max(block_rates)) if block_rates else None},
        "host_ms_per_step": {k: 1e3 * (t_after[k] - t_before[k]) / args.steps for k in t_after},
    }

    final_params = (s.params, np.log(s.weights), np.log(s.lr_weights))   # before the growth run re-uses the context's staging

    # growth trajectory (each NIW config also from init_clusters=1): same data, same context, fresh model
    if not args.no_growth:
        g = host.DPMMSampler(wk, prior, 10.0, N, sampler_seed, burnout=burnout, comm=comm)
        g.init_first_clusters(1)
        fence()
        it, _, lik, kh = g.run_model(args.growth_iters)
        fence()
        tot = float(np.sum(it))
        if dist is not None:
            t = torch.tensor([tot, float(np.sum(it[-25:-5]))], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tot, last = float(t[0]), float(t[1])
        else:
            last = float(np.sum(it[-25:-5]))
        out["growth"] = {"init_clusters": 1, "iterations": args.growth_iters, "it_per_s_whole_run": args.growth_iters / tot,
                         "it_per_s_last20_nonfinal": 20.0 / last, "K_history": [int(k) for k in kh],
                         "log_posterior_final": g.log_posterior()}

    if rank == 0 and not args.no_cpu_baseline and world == 1:
        from oracle import cpu_baseline
        p, logw, loglr = final_params
        inv, _ = host.native.niw_expand(p["R"], want_sigma=False)
        out["cpu_baseline"] = cpu_baseline.run_niw(X, D, K, p["mu"].astype(np.float32), inv.reshape(3 * K, -1).astype(np.float32),
                                                   p["logdet"].astype(np.float32), logw.astype(np.float32), loglr.astype(np.float32), N,
                                                   seconds=args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    wk.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
