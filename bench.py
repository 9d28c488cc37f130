#!/usr/bin/env python3
"""Headline benchmark: Gibbs iterations/sec of the restricted-Gibbs sweep, NIW prior, D=64,
N=10^7 synthetic points in 32 true components, on N GPUs of one node (BASELINE.json metric).

A "step" is one full `group_step` (native master: posterior draws -> parameter hand-over -> fused label +
sub-label sampling kernel -> sort + sufficient statistics kernels -> [all-reduce] -> posterior
update -> split / merge Metropolis steps -> relabel), i.e. exactly the region the reference
times as `iter_count` (src/dp-parallel-sampling.jl:363-366).  One-time work (data generation,
upload, initial labels, burn-in until the split/merge gates are open) is outside the timed
region, as in the reference.

Strong scaling: the N points are fixed and shard over the ranks by contiguous column ranges;
the data-path collective is ONE all-reduce per step inside libdpmmhip.so: 3K packed rows of sufficient statistics (the 2K rows of the
labels as swept + K re-drawn left rows of the clusters a shard reset speculatively; the bad-cluster verdict is read off the reduced rows).

Launch: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no launcher in the environment (WORLD_SIZE unset) the
script starts `python -m torch.distributed.run --nproc-per-node N` on itself as a CHILD process before anything touches the GPU
and exits with the child's code; under the driver's launcher it finds WORLD_SIZE set and runs as a rank.  A run whose world
size is not `--gpus` is an error, never a silent 1-GPU bench.  Prints ONE JSON line on rank 0.

Besides the contract fields the JSON line carries
  roofline      the NIW sweep (for 33 <= D <= 64 two launches: niw_lean_kernel, which finishes the tiles its screens settle and
                dominates, and niw_sweep_direct_kernel<LSTORE,LIST> on the spans it hands on; `launches_ms` splits the time)
                against the matrix pipe: `achieved` = ALGORITHMIC Float32 flops / live duration of the sweep's launches (exceeds the
                Float32 peak because exact cluster screening skips work), `frac` = share of the pipe's time the EXECUTED matrix
                instructions stand for (Float32 ones against the Float32 peak + bf16 ones against the bf16 peak), counted ON THE
                DEVICE in the timed launches (dpmm_last_sweep_work), `dense_*` = the same sweep with screening switched off (every
                cluster evaluated in full) in the same process;
  comm          what the collective saw: world, transport, bytes per all-reduce, all-reduces per step in the timed block (1.0), their HIP-event time;
  blocks        min / median / max it/s over repeated blocks of `--steps` steps (the headline `value` is the first block);
  growth        a run of the same data from ONE initial cluster (`init_clusters=1`): it/s, K history, final log-posterior, NMI;
  legs          (1 GPU only; `--no-legs` skips) short steady-state runs of the other shapes, each with its own roofline entry:
                `overlap_var4` / `overlap_var1` (the headline shape with the component means drawn with MixtureVar 4 and 1
                instead of 100: the middle of the screening range), `inseparable` (the sweep kernel alone on 32 clusters whose means are 0.4 of a
                component's standard deviation apart: no screen can exclude anything -- the worst case, next to the dense leg), `k256` (256 components instead of 32), `c2` (N=10^6), `c3_shard` (what each of
                8 GPUs holds of the headline),
                `c4` (Multinomial D=1000, N=10^6), `c5_shard` (NIW D=256, n=6.25e5), and `shard8_projection`;
  cpu_baseline  the reference algorithm's worker path on the host cores, P worker processes (see oracle/cpu_baseline.py).
"""
import argparse
import gc
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense FP32 matrix peak (no TF32/xf32 on gfx950)
PEAK_BF16_MFMA_TFLOPS = 2516.8  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 = 16 x the FP32 matrix rate (matrix-cores table)
PEAK_HBM_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E ~8 TB/s (6.29 TB/s measured copy)
DATA_SEED, SAMPLER_SEED, BURNOUT, ALPHA = 12345, 123456789, 20, 10.0
WORKER_OPTS = []              # --worker-opt ID=VALUE ...: (option, value) pairs set on every worker of the run


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=float, default=1e7, help="total number of points (default: the BASELINE metric's N)")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--clusters", type=int, default=32)
    ap.add_argument("--blocks", type=int, default=5, help="extra timed blocks of --steps steps after the headline block (min/median/max)")
    ap.add_argument("--settle", type=int, default=100, help="untimed settling sweeps after the burn-in (reported in the line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-growth", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--no-host-master", action="store_true", help="skip the leg with the master's maths on the host")
    ap.add_argument("--no-legs", action="store_true", help="skip the other-shape legs (overlap, c3_shard, c4, c5_shard)")
    ap.add_argument("--legs", default="overlap_var4,overlap_var1,inseparable,k256,c2,c3_shard,c4,c5_shard", help="comma-separated subset of the legs")
    ap.add_argument("--growth-iters", type=int, default=260)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-clock budget of the CPU baseline sample")
    ap.add_argument("--worker-opt", action="append", default=[], metavar="ID=VALUE",
                    help="dpmm_set_option on every worker this run creates (A/B of a library switch on the bench's own legs); reported in config")
    ap.add_argument("--comm-timeout", type=float, default=300.0, help="seconds a rank waits in a collective before it gives up (dead peer)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="development: all ranks on device 0 over gloo + the library's host transport (boxes with one GPU)")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process may use, counted WITHOUT touching HIP (torch.cuda.device_count() falls back to hipGetDeviceCount when amdsmi is
    missing, which initialises the runtime in a parent that is about to start children): the KFD topology's nodes with SIMDs, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when the topology cannot be read -- the ranks' own world / device checks then decide."""
    if not os.path.isdir("/sys/class/kfd"):
        return 0                                 # no amdgpu compute driver on this host at all
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for d in os.listdir(base):
            props = open(os.path.join(base, d, "properties")).read().split("\n")
            kv = dict(l.split(None, 1) for l in props if " " in l)
            if int(kv.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def relaunch_as_ranks(args):
    """--gpus N > 1 without a launcher: start N ranks as a child process (never exec) BEFORE any GPU call and relay its exit code."""
    # (the parent never initialises HIP: the devices are counted from sysfs)
    ndev = visible_gpus()
    if ndev is not None and ndev < args.gpus and not args.share_gpu:
        sys.exit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible on this node (refusing to report a smaller run as n_gpus={args.gpus})")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def kernel_source_tag():
    """sha256 (16 hex digits) over the kernel sources: a committed PMC summary is only quoted while it describes THESE kernels."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dpmmsubclusters.jl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(which, kernel_substr, allow_stale=False):
    """HBM bytes per launch of a kernel from profiles/latest_<which>_pmc_summary.json (written by scripts/collect_profiles.sh together
    with the hash of the kernel sources it profiled): 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, KiB.
    (None, why) when the file is missing or was collected on other kernel sources -- unless allow_stale: then the figure is returned
    with a source text that says on which sources it was collected (the caller reports `traffic_is_current: false` beside it)."""
    f = os.path.join(ROOT, "profiles", f"latest_{which}_pmc_summary.json")
    try:
        pm = json.load(open(f))
        meta = pm.get("_meta", {})
        stale = meta.get("kernel_source_tag") != kernel_source_tag()
        why = f"profiles/latest_{which}_pmc_summary.json is stale (collected on kernel sources {meta.get('kernel_source_tag')})"
        if stale and not allow_stale:
            return None, why
        subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
        tot, hit = 0.0, 0
        for name, c in pm.items():
            if name != "_meta" and any(k in name for k in subs) and "FETCH_SIZE" in c:
                tot += (2.0 * c["FETCH_SIZE"]["median"] + c["WRITE_SIZE"]["median"]) * 1024.0; hit += 1
        if hit:
            return tot, ((why + "; ") if stale else "") + (
                f"profiles/latest_{which}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, median per launch, "
                f"summed over the {hit} kernel(s) of the sweep; {meta.get('collected', '')})")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def pmc_is_current(which):
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", f"latest_{which}_pmc_summary.json")))
        return pm.get("_meta", {}).get("kernel_source_tag") == kernel_source_tag()
    except (OSError, ValueError):
        return False


def pmc_matrix_pipe(which, kernel_substr, launch_ms):
    """What the PMC summary of THIS build says about the matrix pipe in the sweep's kernels (summed over them): busy share
    (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMD-normalised cycles over GRBM_GUI_ACTIVE / 8 XCDs) and the pipe time the counted matrix
    operations stand for: SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 flops against the Float32 peak + SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512
    flops against the bf16 peak, over the live launch time -- the figure `frac` (device-counted matrix instructions) has to agree
    with.  {} when the file is stale or missing."""
    f = os.path.join(ROOT, "profiles", f"latest_{which}_pmc_summary.json")
    try:
        pm = json.load(open(f))
        if pm.get("_meta", {}).get("kernel_source_tag") != kernel_source_tag():
            return {}
        subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
        busy_c = act_c = f32 = bf16 = 0.0
        for name, c in pm.items():
            if name != "_meta" and any(k in name for k in subs) and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                busy_c += c["SQ_VALU_MFMA_BUSY_CYCLES"]["median"]; act_c += c["GRBM_GUI_ACTIVE"]["median"]
                f32 += c["SQ_INSTS_VALU_MFMA_MOPS_F32"]["median"]; bf16 += c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", {}).get("median", 0.0)
        if act_c > 0:
            t = launch_ms * 1e-3 * 1e12
            return {"mfma_busy": (busy_c / 1024.0) / (act_c / 8.0), "pmc_f32_mops_per_launch": f32, "pmc_bf16_mops_per_launch": bf16,
                    "pmc_frac": f32 * 512.0 / t / PEAK_F32_MFMA_TFLOPS + bf16 * 512.0 / t / PEAK_BF16_MFMA_TFLOPS}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        pass
    return {}


# ----------------------------------------------------------------------------------------------- synthetic inputs on the GPU (legs)
def gpu_gaussian_mixture(host, torch, N, D, K, mixture_var, seed):
    """The reference's Gaussian recipe (data_generators.jl:19-42; sizes / means / covariance factors from the product's
    `_mixture_spec`) with the normals drawn on the device: (X torch (N, D) f32 on cuda, labels (N,) int64 host, 1-based)."""
    api = importlib.import_module("dpmmsubclusters_jl_amd.host.api")
    sizes, means, chol = api._mixture_spec(N, D, K, mixture_var, seed)
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    X = torch.randn((N, D), generator=g, device="cuda", dtype=torch.float32)
    a = 0
    for k in range(K):
        b = a + int(sizes[k])
        if b > a:
            Lk = torch.from_numpy(chol[k].astype(np.float32)).cuda()
            X[a:b] = X[a:b] @ Lk.T + torch.from_numpy(means[k].astype(np.float32)).cuda()
        a = b
    return X, np.repeat(np.arange(1, K + 1), sizes).astype(np.int64)


def gpu_multinomial_mixture(torch, N, D, K, trials, seed):
    """The reference's bag-of-words recipe (data_generators.jl:59-72: label ~ U{1..K}; component weights ~ Dir(a), a_d ~ U{1..20} except
    a_k ~ U{30..100}; x ~ Multinomial(trials, p_label)), counts drawn on the device: (X (N, D) f32 cuda, labels host)."""
    rng = np.random.default_rng(seed)
    P = np.zeros((K, D))
    for i in range(K):
        al = rng.integers(1, 21, D).astype(float)
        al[i % D] = rng.integers(30, 101)
        P[i] = rng.dirichlet(al)
    y = rng.integers(1, K + 1, N)
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    Pt = torch.from_numpy(P.astype(np.float32)).cuda()
    yt = torch.from_numpy(y - 1).cuda()
    X = torch.zeros((N, D), device="cuda", dtype=torch.float32)
    ones = torch.ones((1,), device="cuda", dtype=torch.float32)
    step = 100000
    for a in range(0, N, step):
        b = min(N, a + step)
        idx = torch.multinomial(Pt[yt[a:b]], trials, replacement=True, generator=g)
        X[a:b].scatter_add_(1, idx, ones.expand(idx.shape))
    return X, y.astype(np.int64)


def steady_state(pkg, host, torch, prior_kind, prior, X, y, K, steps, settle=30, seed=SAMPLER_SEED, rccl_one_rank=False, engine_opts=()):
    """Upload a device-resident matrix, adopt the generator's labels, burn in until the gates are open, time `steps` group_steps.
    rccl_one_rank: attach a ONE-rank RCCL communicator to the worker first -- every statistics pass then runs the library's collective
    path (occupancy all-reduce, packed-row all-reduce: RCCL's kernels and launches, no wire).  engine_opts: (option, value) pairs."""
    N, D = X.shape
    wk = pkg.Worker(prior_kind, D, N, first_index=0, device=0, seed=seed)
    torch.cuda.synchronize()            # X was produced on torch's stream; the library copies on its own
    wk.upload_points_device(X.data_ptr(), X.stride(0))
    if rccl_one_rank:
        wk.comm_init(wk.comm_unique_id(), 0, 1)
    for opt, val in WORKER_OPTS:
        wk.set_option(opt, val)
    s = host.DPMMSampler(wk, prior, ALPHA, N, seed, burnout=BURNOUT)
    for opt, val in engine_opts:
        s.model.set_option(opt, val)
    s.start_from_labels(y, 1 + np.random.default_rng([DATA_SEED, 7]).integers(0, 2, N), K)
    for _ in range(BURNOUT + 1 + settle):
        s.group_step(False, False)
    wk.sync()
    if prior_kind == pkg.PRIOR_NIW:
        wk.last_sweep_work()
    wk.set_timing(False)                 # the timed loop runs as fit / dp_parallel run it: no timing events between the kernels
    sw, st, ks = [], [], []
    t_before = dict(s.timers)
    # five back-to-back blocks of `steps` steps, the MEDIAN block reported: a leg of 100 steps of 0.4 ms is a 40 ms window, and one host
    # hiccup of a millisecond in it is 2.5 % (the legs ran 3 % slower here than the same shapes through scripts/config_step.py)
    blocks = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            s.group_step(False, False)
        wk.sync()
        blocks.append(time.perf_counter() - t0)
    el = float(np.median(blocks))
    t_after = dict(s.timers)
    work = wk.last_sweep_work() if prior_kind == pkg.PRIOR_NIW else None
    wk.set_timing(True)
    for _ in range(5):                                   # kernel times (HIP events; reading them synchronises): outside the timed loop
        s.group_step(False, False)
        a, b = wk.last_kernel_ms(); sw.append(a); st.append(b); ks.append(s.K)
    lab, _ = wk.get_labels()
    C = np.zeros((int(lab.max()) + 1, K + 1)); np.add.at(C, (lab, y), 1)
    sampler_mod = importlib.import_module("dpmmsubclusters_jl_amd.host.sampler")
    nmi, _ = sampler_mod.nmi_vi_from_contingency(C)
    out = {"n": int(N), "D": int(D), "K_t": float(np.mean(ks)), "steps": steps, "ms_per_step": 1e3 * el / steps, "it_per_s": steps / el,
           "sweep_kernel_ms": float(np.mean(sw)), "stats_kernels_ms": float(np.mean(st)), "nmi_vs_generator": float(nmi),
           "ms_per_step_blocks": [round(1e3 * b / steps, 4) for b in blocks],
           "host_ms_per_step": {k: round(1e3 * (t_after[k] - t_before[k]) / (5 * steps), 4) for k in t_after if t_after[k] - t_before[k] > 0}}
    wk.close()
    return out, work


def inseparable_leg(pkg, torch, n, D, K, sep=0.4, cond=10.0):
    """The screening's worst case as a driver-run number: the sweep kernel on K clusters that no bound separates -- means sep = 0.4 standard
    deviations apart, covariances with a spectrum spread of `cond` in random rotations (the regime of scripts/fuzz_screens.py's cases 5, 10, 17:
    33-46 evaluations per tile of K = 25-42) -- with FIXED parameters (a chain would merge such clusters away), previous labels = the generating
    components, bin-sorted visiting order.  Reports the sweep kernel's launch time (HIP events), evaluations per tile and the executed fraction."""
    rng = np.random.default_rng([DATA_SEED, 11])
    mus = rng.normal(size=(3 * K, D)) * sep
    for k in range(K):
        d = rng.normal(size=D) * 0.4
        mus[3 * k + 1] = mus[3 * k] + d; mus[3 * k + 2] = mus[3 * k] - d
    Sig = np.empty((3 * K, D, D))
    for j in range(3 * K):
        Q, _ = np.linalg.qr(rng.normal(size=(D, D)))
        ev = np.exp(rng.uniform(-0.5 * np.log(cond), 0.5 * np.log(cond), D))
        Sig[j] = (Q * ev) @ Q.T
    invS = np.linalg.inv(Sig); invS = 0.5 * (invS + invS.transpose(0, 2, 1))
    logdet = np.linalg.slogdet(Sig)[1]
    sizes = np.full(K, n // K); sizes[: n - sizes.sum()] += 1
    g = torch.Generator(device="cuda"); g.manual_seed(DATA_SEED + 11)
    X = torch.randn((n, D), generator=g, device="cuda", dtype=torch.float32)
    a = 0
    for k in range(K):
        b = a + int(sizes[k])
        Lk = torch.from_numpy(np.linalg.cholesky(Sig[3 * k]).astype(np.float32)).cuda()
        X[a:b] = X[a:b] @ Lk.T + torch.from_numpy(mus[3 * k].astype(np.float32)).cuda()
        a = b
    z = np.repeat(np.arange(1, K + 1), sizes).astype(np.int64)
    w = np.full(K, 1.0 / K, np.float32); lr = np.full((K, 2), 0.5, np.float32)
    par = (mus.astype(np.float32), invS.reshape(3 * K, -1).astype(np.float32), logdet.astype(np.float32), lr, w)
    wk = pkg.Worker(pkg.PRIOR_NIW, D, n, first_index=0, device=0, seed=SAMPLER_SEED)
    torch.cuda.synchronize()
    wk.upload_points_device(X.data_ptr(), X.stride(0))
    for opt, val in WORKER_OPTS:
        wk.set_option(opt, val)
    wk.set_params_niw(*par)
    wk.set_labels(z, 1 + (np.arange(n) & 1))
    wk.suffstats_packed(None)
    wk.set_timing(True)
    ms = []
    for ep in range(1, 6):
        wk.set_params_niw(*par)
        if ep == 2:
            wk.last_sweep_work()                  # (counted from the second sweep on: previous labels = the kernel's own draws)
        wk.sweep(ep); wk.sync()
        ms.append(wk.last_kernel_ms()[0])
        wk.suffstats_packed(None)
    work = wk.last_sweep_work()
    lab, _ = wk.get_labels()
    wk.close()
    sweep_ms = float(np.median(ms[1:]))
    return {"n": int(n), "D": int(D), "K_t": float(K), "sweep_kernel_ms": sweep_ms, "sweep_kernel_ms_all": [round(v, 3) for v in ms],
            "labels_kept_fraction": float((lab == z).mean()),
            "workload": f"sweep kernel alone, NIW D={D} N={n:g} K={K} FIXED parameters: component means {sep} sd apart, covariance spectra spread {cond:g} in random rotations "
                        "(inseparable clusters: no screen excludes anything)",
            "roofline": niw_roofline(n, D, float(K), sweep_ms, work)}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


SWEEP_KERNELS_64 = ("niw_lean_kernel", "niw_sweep_direct_kernel", "niw_sub_kernel")


def sweep_kernel_names(D):
    """The launches of one NIW sweep.  33 <= D <= 64: niw_lean_kernel (every tile; finishes the tiles whose label candidates the
    screens settle) and niw_sweep_direct_kernel<..., LSTORE, LIST> (labels and sub-labels of the spans it handed on); without the lean
    launch (overlapping clusters; K > 64 on the set_params path, whose pre-screen runs in the sweep kernel) niw_sweep_direct_kernel<..., LSTORE> (labels) and niw_sub_kernel (sub-labels) on every tile;
    D <= 32: niw_sweep_direct_kernel alone; D > 64: niw_sweep_kernel (behind its bracket launch)."""
    return "+".join(SWEEP_KERNELS_64) if 32 < D <= 64 else ("niw_sweep_direct_kernel" if D <= 32 else "niw_sweep_kernel")


def pipe_frac(work, sweep_ms):
    """Share of the matrix pipe's time the device-counted matrix instructions of the sweep stand for: Float32 instructions at the
    Float32 rate + bf16 instructions at the bf16 rate (one pipe executes both), over the live duration of the sweep's launches."""
    t = sweep_ms * 1e-3 * 1e12
    return work["executed_flops"] / t / PEAK_F32_MFMA_TFLOPS + work["bf16_flops"] / t / PEAK_BF16_MFMA_TFLOPS


def niw_roofline(n, D, k_mean, sweep_ms, work):
    flops_alg = 2.0 * n * D * D * (k_mean + 2)
    exe = work["executed_flops"]
    return {"kernel": sweep_kernel_names(D), "bound": "mfma",
            "achieved": flops_alg / (sweep_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": pipe_frac(work, sweep_ms), "f32_frac": exe / (sweep_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "traffic": None, "avg_launch_ms": sweep_ms, "algorithmic_bytes_per_launch": (4.0 * D + 4.0) * n,
            "hbm_frac": (4.0 * D + 4.0) * n / (sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
            "algorithmic_flops_per_launch": flops_alg, "executed_flops_per_launch": exe, "executed_bf16_flops_per_launch": work["bf16_flops"],
            "pruning_factor": flops_alg / exe if exe else None, "b3_evals_per_wave_tile": work.get("b3_evals", 0.0) / max(1.0, work["wave_tiles"]),
            "full_evals_per_wave_tile": work["full_evals"] / max(1.0, work["wave_tiles"]),
            "screens16_per_wave_tile": work["screens16"] / max(1.0, work["wave_tiles"]),
            "tail_pairs_per_wave_tile": work["tail_pairs"] / max(1.0, work["wave_tiles"]),
            "brackets_per_wave_tile": work["brackets"] / max(1.0, work["wave_tiles"]),
            "bf16_bottom_screens_per_wave_tile": work["bf16_bottom_screens"] / max(1.0, work["wave_tiles"]),
            "bf16_top_screens_per_wave_tile": work["bf16_top_screens"] / max(1.0, work["wave_tiles"]),
            "direction_screens_per_wave_tile": work.get("direction_screens", 0.0) / max(1.0, work["wave_tiles"]),
            "bf16_mfma_per_tile": work["bf16_mfma"] / max(1.0, work["wave_tiles"])}


def run_legs(args, pkg, host, torch, one_gpu_ms):
    legs, want = {}, [x for x in args.legs.split(",") if x]
    D, K = 64, 32
    niw64 = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    for name, var in (("overlap_var4", 4.0), ("overlap_var1", 1.0)):
        if name not in want:
            continue
        X, y = gpu_gaussian_mixture(host, torch, 10 ** 7, D, K, var, DATA_SEED)
        r, work = steady_state(pkg, host, torch, pkg.PRIOR_NIW, niw64, X, y, K, 20)
        r["workload"] = f"headline shape (NIW D=64 N=1e7 K=32) with component means ~ N(0, {var:g} I) instead of N(0, 100 I); from the generator's labels"
        r["roofline"] = niw_roofline(r["n"], D, r["K_t"], r["sweep_kernel_ms"], work)
        legs[name] = r
        del X
        torch.cuda.empty_cache()
    if "inseparable" in want:
        legs["inseparable"] = inseparable_leg(pkg, torch, 10 ** 7, D, K)
        torch.cuda.empty_cache()
    if "k256" in want:
        Kb = 256
        X, y = gpu_gaussian_mixture(host, torch, 10 ** 7, D, Kb, 100.0, DATA_SEED)
        r, work = steady_state(pkg, host, torch, pkg.PRIOR_NIW, niw64, X, y, Kb, 20)
        r["workload"] = "headline shape with 256 true components (NIW D=64 N=1e7, MixtureVar 100): K_t = 256 live clusters, 32 640 merge pairs per step"
        r["roofline"] = niw_roofline(r["n"], D, r["K_t"], r["sweep_kernel_ms"], work)
        legs["k256"] = r
        del X
        torch.cuda.empty_cache()
    if "c2" in want:
        n = 10 ** 6
        X, y = gpu_gaussian_mixture(host, torch, n, D, K, 100.0, DATA_SEED)
        r, work = steady_state(pkg, host, torch, pkg.PRIOR_NIW, niw64, X, y, K, 100, settle=60)
        r["workload"] = "C2: NIW D=64, N=1e6, 32 true components, one GPU (steady state from the generator's labels)"
        r["roofline"] = niw_roofline(n, D, r["K_t"], r["sweep_kernel_ms"], work)
        legs["c2"] = r
        del X
        torch.cuda.empty_cache()
    if "c3_shard" in want:
        n = 1250000
        X, y = gpu_gaussian_mixture(host, torch, n, D, K, 100.0, DATA_SEED)
        r, work = steady_state(pkg, host, torch, pkg.PRIOR_NIW, niw64, X, y, K, 100, settle=60)
        r["workload"] = "what each of 8 GPUs holds of the headline: NIW D=64, n=1.25e6, K=32, one GPU, no collective"
        r["roofline"] = niw_roofline(n, D, r["K_t"], r["sweep_kernel_ms"], work)
        tr, src = pmc_traffic("shard", SWEEP_KERNELS_64, allow_stale=True)
        r["roofline"].update({"traffic": tr, "traffic_source": src, "traffic_is_current": pmc_is_current("shard") if tr else None})
        legs["c3_shard"] = r
        # 1 -> 8 projection.  A rank of the 8-GPU run does this step in the ONE-COLLECTIVE form of the per-step pass (speculative reset, 3K
        # rows through one all-reduce, finalize kernel: dpmm_api.cpp run_stats) plus that all-reduce over xGMI.  The form is measured here
        # with a one-rank RCCL communicator attached; the all-reduce is not measurable on one GPU and stays an ASSUMPTION, stated as such.
        try:
            r1, _ = steady_state(pkg, host, torch, pkg.PRIOR_NIW, niw64, X, y, K, 100, settle=60, rccl_one_rank=True)
            rccl_ms, rccl_err = r1["ms_per_step"], None
        except Exception as e:      # RCCL not loadable on this box: say so, keep the assumption
            rccl_ms, rccl_err = None, repr(e)[:200]
        # (RCCL elides an in-place all-reduce of ONE rank: what the one-rank leg adds to the step is the pass's own extra kernels -- the finalize
        # kernel behind the all-reduce, the undo of the speculative reset -- not the collective.  The collective itself stays an assumption.)
        assumed_allreduce = 0.04
        with_comm = (rccl_ms if rccl_ms is not None else r["ms_per_step"]) + assumed_allreduce
        legs["shard8_projection"] = {"one_gpu_ms_per_step": one_gpu_ms, "shard_ms_per_step": r["ms_per_step"],
                                     "shard_ms_per_step_one_collective_form": rccl_ms, "rccl_1rank_error": rccl_err,
                                     "measured_extra_kernels_ms_per_step": (rccl_ms - r["ms_per_step"]) if rccl_ms is not None else None,
                                     "assumed_allreduce_ms_per_step": assumed_allreduce,
                                     "collectives_per_step": 1,
                                     "projected_speedup_1_to_8": one_gpu_ms / with_comm,
                                     "speedup_without_collectives": one_gpu_ms / r["ms_per_step"],
                                     "target_shard_plus_collectives_ms_for_6x": one_gpu_ms / 6.0,
                                     "note": "projection from one GPU.  Measured: the shard's step, and the same step in the one-collective form of a multi-rank "
                                             "run (one-rank RCCL communicator: RCCL elides the all-reduce itself).  Assumed: 0.04 ms for the ONE all-reduce of "
                                             "3K packed rows (1.65 MB) over xGMI incl. the wait for the slowest rank; the driver's SCALE run decides"}
        del X
        torch.cuda.empty_cache()
    if "c4" in want:
        n, Dm = 10 ** 6, 1000
        X, y = gpu_multinomial_mixture(torch, n, Dm, K, 100, DATA_SEED)
        prior = host.multinomial_hyper(np.ones(Dm, np.float32))          # test/save_load_test/multinomial_params.jl:24
        r, _ = steady_state(pkg, host, torch, pkg.PRIOR_MULT, prior, X, y, K, 50)
        r["workload"] = "C4: Multinomial D=1000 N=1e6 K=32 (100 trials per point), one GPU"
        # Bytes: SURVEY 8d's per-unit figure is for Float32 points (4 n D + 16 n); the context keeps count data as a LOSSLESS byte copy
        # ([n][roundup(D, 128)] u8) and every Multinomial kernel streams that, so the bytes the dominant kernel has to move are
        # n (ld8 + 4 [visiting order] + 4 [previous bin] + 4 [new bin]).  `frac` prices THOSE against the kernel's own launch time (it cannot
        # exceed 1); the Float32 figure over the whole pass is kept as `survey_*` and the counter traffic as `traffic_frac`.
        ld8 = (Dm + 127) // 128 * 128
        bytes_f32 = 4.0 * n * Dm + 16.0 * n
        bytes_u8 = float(n) * (ld8 + 12.0)
        t_sweep = r["sweep_kernel_ms"] * 1e-3
        t = (r["sweep_kernel_ms"] + r["stats_kernels_ms"]) * 1e-3
        tr_sweep, src = pmc_traffic("mult", "mult_sweep", allow_stale=True)
        tr_stats, _ = pmc_traffic("mult", "mult_stats", allow_stale=True)
        r["roofline"] = {"kernel": "mult_sweep_u8_kernel", "bound": "hbm",
                         "achieved": bytes_u8 / t_sweep / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": bytes_u8 / t_sweep / 1e9 / PEAK_HBM_GBPS,
                         "algorithmic_bytes_per_launch": bytes_u8, "avg_launch_ms": r["sweep_kernel_ms"],
                         "traffic": tr_sweep, "traffic_source": src, "traffic_is_current": pmc_is_current("mult") if tr_sweep else None,
                         "traffic_frac": (tr_sweep / t_sweep / 1e9 / PEAK_HBM_GBPS) if tr_sweep else None,
                         "pass_traffic": (tr_sweep + tr_stats) if (tr_sweep and tr_stats) else None,
                         "pass_traffic_frac": ((tr_sweep + tr_stats) / t / 1e9 / PEAK_HBM_GBPS) if (tr_sweep and tr_stats) else None,
                         "survey_float32_bytes_per_step": bytes_f32, "survey_GBps_sweep_plus_stats": bytes_f32 / t / 1e9,
                         "survey_frac_sweep_plus_stats": bytes_f32 / t / 1e9 / PEAK_HBM_GBPS}
        legs["c4"] = r
        del X
        torch.cuda.empty_cache()
    if "c5_shard" in want:
        n, Dh = 625000, 256
        X, y = gpu_gaussian_mixture(host, torch, n, Dh, K, 100.0, DATA_SEED)
        prior = host.niw_hyperparams(1.0, np.zeros(Dh), Dh + 3, np.eye(Dh))
        r, work = steady_state(pkg, host, torch, pkg.PRIOR_NIW, prior, X, y, K, 30)
        r["workload"] = "what each of 8 GPUs holds of C5: NIW D=256, n=6.25e5, K=32, one GPU, no collective"
        r["roofline"] = niw_roofline(n, Dh, r["K_t"], r["sweep_kernel_ms"], work)
        tr, src = pmc_traffic("d256", ("niw_sweep_kernel", "niw_bracket_big_kernel"), allow_stale=True)
        r["roofline"].update({"traffic": tr, "traffic_source": src, "traffic_is_current": pmc_is_current("d256") if tr else None})
        legs["c5_shard"] = r
        del X
        torch.cuda.empty_cache()
    return legs

COMPACT_LIMIT = 4096          # bytes of the ONE stdout line (the driver keeps an 8 KB tail and parses the last line)


def compact_line(out):
    """The ONE JSON line of the contract (last line of stdout), < 4 KB whatever the run measured: the contract fields, `config`
    (workload, master, also_measured: scalars only), `roofline` of the dominant kernel and `cpu_baseline`.  Everything else the run
    measured (legs, growth history, host-master block, comm, blocks, the full roofline with its definitions) is `bench_details.json`
    beside this file and a dump on stderr.  Round 5's line had grown to 20 KB and the driver could not parse it."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out[k] for k in keep if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "points_per_gpu", "parallelism", "master", "worker_options_overridden", "also_measured") if cfg.get(k) is not None}
    r = out.get("roofline", {})
    rk = ("kernel", "bound", "achieved", "peak", "unit", "frac", "hbm_frac", "traffic", "traffic_frac", "traffic_is_current", "algorithmic_bytes_per_launch",
          "avg_launch_ms", "lean_kernel_ms", "mfma_pipe_frac", "pmc_frac", "mfma_busy", "dense_f32_frac", "dense_launch_ms", "pruning_factor", "stats_kernels_ms")
    line["roofline"] = {k: r[k] for k in rk if r.get(k) is not None or k == "traffic"}
    c = out.get("cpu_baseline")
    if c is not None:
        line["cpu_baseline"] = {k: c[k] for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "julia_found") if k in c}
    cm = out.get("comm")
    if cm and cm.get("world", 1) > 1:          # what the collective saw, for the driver's scaling record
        line["comm"] = {"world": cm["world"], "transport": cm["transport"], "rows_allreduce_bytes": cm["rows_allreduce_bytes"],
                        "allreduces_per_step": cm["allreduces_per_step_in_timed_block"], "rows_allreduce_ms": cm["rows_allreduce_ms"],
                        "one_collective_per_step_pass": cm["one_collective_per_step_pass"]}
    line["details"] = "bench_details.json (written beside bench.py; also dumped on stderr)"

    def rnd(o):
        if isinstance(o, float):
            return float(f"{o:.5g}")
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [rnd(v) for v in o]
        return o
    line = rnd(line)
    line["value"] = out["value"]; line["ms_per_step"] = out["ms_per_step"]     # the headline itself at full precision
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > COMPACT_LIMIT:          # never let prose push the line over: drop the optional texts, longest first
        for path in (("config", "master"), ("cpu_baseline", "sample"), ("config", "parallelism")):
            d = line.get(path[0], {})
            if path[1] in d and len(txt) > COMPACT_LIMIT:
                d[path[1]] = str(d[path[1]])[:160]
                txt = json.dumps(line, separators=(",", ":"))
    assert len(txt) <= COMPACT_LIMIT and "\n" not in txt, len(txt)
    return txt


def emit(out):
    """Write the full record to bench_details.json + stderr, then the compact contract line as the LAST line of stdout."""
    full = json.dumps(out)
    try:
        with open(os.path.join(ROOT, "bench_details.json"), "w") as f:
            f.write(full + "\n")
    except OSError as e:
        print(f"bench.py: bench_details.json not written ({e})", file=sys.stderr)
    print("bench.py details: " + full, file=sys.stderr, flush=True)
    sys.stdout.flush()
    try:       # C stdio of the libraries in this process (RCCL prints its version banner through a buffered printf that otherwise lands BEHIND the line at exit)
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    print(compact_line(out), flush=True)


def main():
    args = parse_args()
    WORKER_OPTS.extend((int(kv.split("=")[0]), float(kv.split("=")[1])) for kv in args.worker_opt)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_as_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus} (launch `--nproc-per-node {args.gpus}`, or call bench.py without a launcher)")
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        # a rank that dies must end the run with a non-zero exit, not hang it: bounded waits in the process group (host transport: gloo's
        # all_reduce raises -> the library's callback fails -> DPMM_ECOMM) and in the library (RCCL: DPMM_OPT_COMM_TIMEOUT_MS, set below)
        import datetime
        tmo = datetime.timedelta(seconds=args.comm_timeout)
        if args.share_gpu:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=tmo)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")

    from __graft_entry__ import load_package
    pkg = load_package()
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    from dpmmsubclusters_jl_amd.host.comm import TorchDistComm
    from dpmmsubclusters_jl_amd import binding

    N, D, K = int(args.points), args.dim, args.clusters
    comm = TorchDistComm(device=local_rank) if world > 1 else host.LocalComm()
    lo, hi = (N * rank) // world, (N * (rank + 1)) // world
    X, y = host.gaussian_mixture_shard(N, D, K, 100.0, DATA_SEED, lo, hi)

    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))      # default prior, dp-parallel-sampling.jl:272-274
    wk = pkg.Worker(pkg.PRIOR_NIW, D, hi - lo, first_index=lo, device=local_rank, seed=SAMPLER_SEED)
    wk.upload_points(X)
    wk.set_option(binding.OPT_COMM_TIMEOUT_MS, 1e3 * args.comm_timeout)
    for opt, val in WORKER_OPTS:
        wk.set_option(opt, val)
    s = host.DPMMSampler(wk, prior, ALPHA, N, SAMPLER_SEED, burnout=BURNOUT, comm=comm)
    sub0 = 1 + (np.random.default_rng([DATA_SEED, 7, rank]).integers(0, 2, hi - lo))
    s.start_from_labels(y, sub0, K)
    # burn-in (setup, untimed): `burnout` sweeps until every cluster's split/merge gate is open
    for _ in range(BURNOUT + 1):
        s.group_step(False, False)
    # ... and `--settle` more (setup, untimed; reported as `settle`): the first ~100 steps of a process run ~2 % slower than all later ones
    # (clock / power state of a GPU that was idle during the upload) -- the metric is the steady-state rate of a long run
    for _ in range(args.settle):
        s.group_step(False, False)
    # (the W warm-up steps run further down: directly in front of the timed block, behind everything else the block needs set up)

    def fence():
        torch.cuda.synchronize()
        wk.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(vals):
        if dist is None:
            return [float(v) for v in vals]
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if args.share_gpu else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def timed_block(nsteps, collect=None):
        # No cyclic garbage collection INSIDE a timed block -- and no collection in front of it either: gc.collect() here kept the host busy for a
        # few hundred milliseconds, the GPU idled, its clocks fell back, and the 45 ms block that followed ran its kernels 6 % slower (measured on
        # one box, alternating: 640 / 647 it/s with the collect, 682 / 685 without; sweep launch 1.02 against 0.95 ms).  The warm-up steps run
        # right up to the block for the same reason.
        gc.disable()
        try:
            fence()
            t0 = time.perf_counter()
            for _ in range(nsteps):
                s.group_step(False, False)
                if collect is not None:
                    collect()
            fence()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        return max_over_ranks([dt])[0]

    n_local = hi - lo
    sweep_ms, stats_ms, ks, work, comm_ms = [], [], [], [], []

    def collect():     # HIP events on the library's stream (the stream is idle here: stats were read back)
        a, _ = wk.last_kernel_ms()
        sweep_ms.append(a); ks.append(s.K)

    parts_ms = []

    def collect_all():
        _, b = wk.last_kernel_ms()
        stats_ms.append(b)
        parts_ms.append(wk.last_sweep_parts_ms())
        if world > 1:
            comm_ms.append(wk.last_comm_ms())

    # The headline block carries the events around the dominant kernel only (two barrier packets per step: its launch duration is
    # measured live in the timed region); the statistics / all-reduce events (four to eight more per step) are recorded in the first
    # of the extra blocks.  fit / dp_parallel record none (DPMM_OPT_KERNEL_TIMING is off by default).
    wk.set_timing(1)
    # The W untimed warm-up steps, as the timed ones will run (same timing mode, same per-step event read), with NOTHING between them and the
    # block but its fence: a host pause there -- a garbage collection, the set-up calls below on a bad day -- lets the GPU idle, its clocks fall
    # back, and a 45 ms block runs 5-20 % slower (one default run of this round: 544 it/s with blocks 652-674 behind it; another: 646 with 663-692).
    # (so the host timers, the device's work counters and the collective counts below cover the W warm-up steps + the K timed ones: per-step / per-launch
    #  averages of steps that are all alike)
    gc.disable()
    t_before = dict(s.timers)
    wk.last_sweep_work()              # clear the device's work counters: they add up over the launches that follow and are read once afterwards
    ci0 = wk.comm_info()
    for _ in range(args.warmup):
        s.group_step(False, False)
        wk.last_kernel_ms()
    elapsed = timed_block(args.steps, collect)
    ci1 = wk.comm_info()
    work.append(wk.last_sweep_work())     # per-launch averages over exactly the timed launches
    t_after = dict(s.timers)
    wk.set_timing(15)
    timed_block(args.steps, collect_all)
    wk.set_timing(1)
    block_rates = []
    for _ in range(max(0, args.blocks)):
        block_rates.append(args.steps / timed_block(args.steps))
    wk.set_timing(7)

    k_mean = float(np.mean(ks))
    flops_alg = 2.0 * n_local * D * D * (k_mean + 2)       # likelihood vs K clusters + own left/right (per point x points)
    avg_sweep_ms = float(np.mean(sweep_ms))
    exe = float(np.mean([w["executed_flops"] for w in work]))
    achieved = flops_alg / (avg_sweep_ms * 1e-3) / 1e12
    # HBM bytes per launch of the sweep kernel: PMC counters need rocprofv3, so the figure comes from the counter summary that
    # scripts/collect_profiles.sh wrote for THIS command (profiles/latest_bench_pmc_summary.json); it carries the hash of the kernel
    # sources it was collected on and is not quoted for any other build or configuration
    sweep_kernels = SWEEP_KERNELS_64 if 32 < D <= 64 else ("niw_sweep_direct_kernel" if D <= 32 else "niw_sweep_kernel",)
    headline_shape = (N == 10 ** 7 and D == 64 and world == 1)
    traffic, traffic_source = (pmc_traffic("bench", sweep_kernels, allow_stale=True) if headline_shape else (None, None))
    exe_bf16 = float(np.mean([w["bf16_flops"] for w in work]))
    pm = np.asarray(parts_ms, np.float64).mean(axis=0) if parts_ms else np.zeros(3)
    # Which roof: the sweep's Float32 flops of SURVEY 8d are not what the kernel executes any more (exact screening removes 31 of 32
    # clusters per tile on this data: algorithmic flops / time is 16 x the Float32 matrix peak), so the ceiling that bounds the launch is
    # HBM: every point's D features are read once per sweep whatever the screens decide.  `achieved` = SURVEY 8d's bytes per point
    # (4 D + 4) x the points of the launch / live launch time; the matrix pipe's share is kept beside it (`mfma_pipe_frac`).
    alg_bytes = 4.0 * n_local * D + 4.0 * n_local
    hbm_gbps = alg_bytes / (avg_sweep_ms * 1e-3) / 1e9
    roof = {"kernel": sweep_kernel_names(D), "bound": "hbm",
            "achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm_gbps / PEAK_HBM_GBPS, "hbm_frac": hbm_gbps / PEAK_HBM_GBPS,
            "traffic": traffic, "traffic_source": traffic_source, "traffic_is_current": (pmc_is_current("bench") if traffic is not None else None),
            "traffic_frac": (traffic / (avg_sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS) if traffic else None,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_sweep_ms, "lean_kernel_ms": float(pm[0]) if parts_ms else None,
            "mfma_pipe_frac": float(np.mean([pipe_frac(w, avg_sweep_ms) for w in work])),
            "f32_frac": exe / (avg_sweep_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "bf16_frac": exe_bf16 / (avg_sweep_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, "peak_f32_mfma": PEAK_F32_MFMA_TFLOPS, "peak_bf16": PEAK_BF16_MFMA_TFLOPS,
            "executed_bf16_flops_per_launch": exe_bf16,
            "launches_ms": {"niw_lean_kernel": float(pm[0]),
                            "niw_sweep_direct_kernel<LSTORE,LIST> (labels and sub-labels of the spans handed on)": float(pm[1]),
                            "niw_sub_kernel (every tile's sub-labels: only in sweeps without the lean launch)": float(pm[2]),
                            "note": "HIP events between the launches of one sweep, recorded in the second block (timing bit 8); "
                                    "avg_launch_ms is the sweep's events of the headline block"},
            "algorithmic_flops_per_launch": flops_alg, "executed_flops_per_launch": exe,
            "executed_tflops": exe / (avg_sweep_ms * 1e-3) / 1e12, "pruning_factor": flops_alg / exe if exe else None,
            "algorithmic_tflops": achieved, "algorithmic_mfma_frac": achieved / PEAK_F32_MFMA_TFLOPS,
            "work_per_launch": {k: float(np.mean([w[k] for w in work])) for k in ("wave_tiles", "full_evals", "screens16", "tail_pairs", "brackets")},
            "bf16_mfma_per_tile": float(np.mean([w["bf16_mfma"] / max(1.0, w["wave_tiles"]) for w in work])),
            "frac_definition": "`frac` = `hbm_frac` = algorithmic bytes of the sweep (SURVEY 8d: (4 D + 4) bytes per point) / live duration of the sweep's "
                               "launches / 8 TB/s.  `mfma_pipe_frac` = share of the matrix pipe's time: matrix instructions counted on the device in the "
                               "timed launches, Float32 ones x 2048 flops against the Float32 peak (`f32_frac`) + bf16 ones x 16384 flops against the "
                               "bf16 peak (`bf16_frac`: reference brackets, bf16 screens, three-plane sub-cluster evaluations) "
                               "(= (SQ_INSTS_VALU_MFMA_MOPS_F32 / peak_f32 + ..._BF16 / peak_bf16) x 512: `pmc_frac`); `algorithmic_tflops` = the "
                               "sweep's algorithmic Float32 flops (SURVEY 8d) over the same duration: above the Float32 peak because exact screening "
                               "skips clusters; `dense_*` = the same sweep with screening off",
            "stats_kernels_ms": float(np.mean(stats_ms)), "kernel_source_tag": kernel_source_tag()}
    if headline_shape:
        roof.update(pmc_matrix_pipe("bench", sweep_kernels, avg_sweep_ms))

    # same kernel, same process, screening off: every cluster is evaluated in full (labels are bit-identical by construction)
    if not args.no_dense:
        wk.set_option(binding.OPT_SCREEN_MARGIN, 0.0)
        dm, dw = [], []
        for _ in range(3):
            s.group_step(False, False)
            dm.append(wk.last_kernel_ms()[0]); dw.append(wk.last_sweep_work())
        wk.set_option(binding.OPT_SCREEN_MARGIN, 50.0)
        d_ms, d_fl = float(np.mean(dm[1:])), float(np.mean([w["executed_flops"] for w in dw[1:]]))
        roof.update({"dense_launch_ms": d_ms, "dense_executed_tflops": d_fl / (d_ms * 1e-3) / 1e12,
                     "dense_frac": float(np.mean([pipe_frac(w, d_ms) for w in dw[1:]])),
                     "dense_f32_frac": d_fl / (d_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "dense_algorithmic_tflops": flops_alg / (d_ms * 1e-3) / 1e12})

    info = wk.comm_info()
    out = {
        "metric": "Gibbs iterations/sec, N=10M D=64 NIW" if (N == 10 ** 7 and D == 64) else f"Gibbs iterations/sec, N={N} D={D} NIW",
        "value": args.steps / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle": args.settle,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"NIW D={D} N={N} synthetic GMM of the reference's generator (data_generators.jl:19-42), {K} true components, "
                               f"MixtureVar 100 (component means ~ N(0, 100 I): well-separated clusters -- overlapping ones are the overlap_* entries "
                               f"of also_measured), K_t={k_mean:.1f} live clusters, alpha=10, default NIW prior, steady state after {BURNOUT + 1} burn-in "
                               f"+ {args.settle} settling sweeps",
                   "points_per_gpu": n_local, "worker_options_overridden": {str(o): v for o, v in WORKER_OPTS} or None,
                   "parallelism": f"points sharded over {world} GPU(s); all-reduces of a statistics pass inside libdpmmhip.so: see comm",
                   "master": "DEVIATION from north_star's wording: the headline runs the engine's default for D >= 64 -- posteriors, factorisations and "
                             "parameter draws on the GPU beside the statistics (DPMMH_OPT_DEVICE_MASTER), every decision (gates, split / merge Metropolis "
                             "steps) on the host.  north_star's configuration (draws on the host) = also_measured.host_master_*"},
        "roofline": roof,
        "comm": {"world": info["world"], "transport": info["transport"], "occupancy_allreduce_bytes": info["counts_bytes"],
                 "rows_allreduce_bytes": info["rows_bytes"], "allreduces_since_attach": info["allreduces"],
                 "one_collective_per_step_pass": info["one_collective"],
                 "allreduces_per_step_in_timed_block": (ci1["allreduces"] - ci0["allreduces"]) / (args.steps + args.warmup),
                 "occupancy_allreduce_ms": float(np.mean([c[0] for c in comm_ms])) if comm_ms else None,
                 "rows_allreduce_ms": float(np.mean([c[1] for c in comm_ms])) if comm_ms else None,
                 "note": "HIP events on the ctx stream around each all-reduce of the timed steps (rank 0); they include waiting for the slowest rank"},
        "blocks": {"it_per_s": block_rates, "min": float(np.min(block_rates)) if block_rates else None,
                   "median": float(np.median(block_rates)) if block_rates else None, "max": float(np.max(block_rates)) if block_rates else None},
        "host_ms_per_step": {k: 1e3 * (t_after[k] - t_before[k]) / (args.steps + args.warmup) for k in t_after},
    }

    final_params = (s.params, np.log(s.weights), np.log(s.lr_weights))   # before the growth run re-uses the context's staging

    # growth trajectory (each NIW config also from init_clusters=1): same data, same context, fresh model
    if not args.no_growth:
        g = host.DPMMSampler(wk, prior, ALPHA, N, SAMPLER_SEED, burnout=BURNOUT, comm=comm)
        g.init_first_clusters(1)
        fence()
        it, nmi, _, kh = g.run_model(args.growth_iters, gt=None)
        fence()
        tot, last = max_over_ranks([float(np.sum(it)), float(np.sum(it[-25:-5]))])
        wk.set_ground_truth_range(y - 1, K)
        gn, _ = importlib.import_module("dpmmsubclusters_jl_amd.host.sampler").nmi_vi_from_contingency(comm.reduce_counts(wk.contingency(g.K)))
        out["growth"] = {"init_clusters": 1, "iterations": args.growth_iters, "it_per_s_whole_run": args.growth_iters / tot,
                         "it_per_s_last20_nonfinal": 20.0 / last, "K_history": [int(k) for k in kh], "K_final": int(kh[-1]), "K_true": K,
                         "log_posterior_final": g.log_posterior(), "nmi_final_vs_generator": float(gn)}
        # labels MOVING: the 40 iterations from iteration 100 of this run (K still growing, splits accepted, whole clusters relabelled: the
        # cached cluster rows of the derived statistics are invalidated every few steps) -- the other end of the frozen-label steady state
        if args.growth_iters >= 140:
            mid = max_over_ranks([float(np.sum(it[100:140]))])[0]
            out["growth"]["moving_labels"] = {"iterations": "100..139", "it_per_s": 40.0 / mid, "K_at_100": int(kh[100]), "K_at_139": int(kh[139]),
                                              "K_changes": int(np.count_nonzero(np.diff(np.asarray(kh[100:140]))))}

    # the configuration north_star words: posterior parameter draws and split / merge steps on the HOST (DPMMH_OPT_DEVICE_MASTER = 0) --
    # same data, same context, same steady state; the headline runs the engine's default for D >= 64 (device master, dpmm_hip_master.h)
    if not args.no_host_master:
        engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
        hm = host.DPMMSampler(wk, prior, ALPHA, N, SAMPLER_SEED, burnout=BURNOUT, comm=comm)
        hm.model.set_option(engine.OPT_DEVICE_MASTER, 0)
        hm.start_from_labels(y, sub0, K)
        for _ in range(BURNOUT + 1 + 30):
            hm.group_step(False, False)
        wk.set_timing(0)
        hb = dict(hm.timers)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hm.group_step(False, False)
        fence()
        el = max_over_ranks([time.perf_counter() - t0])[0]
        ha = dict(hm.timers)
        wk.set_timing(7)
        out["host_master"] = {"it_per_s": args.steps / el, "ms_per_step": 1e3 * el / args.steps, "K_t": int(hm.K),
                              "ratio_to_headline": (args.steps / el) / out["value"], "cpu_model": _cpu_model(), "host_threads": int(hm.nthreads),
                              "host_ms_per_step": {k: round(1e3 * (ha[k] - hb[k]) / args.steps, 4) for k in ha if ha[k] - hb[k] > 0},
                              "note": "DPMMH_OPT_DEVICE_MASTER = 0: posteriors, factorisations, parameter draws and every Metropolis step on the host "
                                      "(src/shared_actions.jl:41-66, src/priors/niw.jl:20-40), parameters through dpmm_params_staging / dpmm_commit_params"}

    if rank == 0 and world == 1 and not args.no_legs:
        out["legs"] = run_legs(args, pkg, host, torch, out["ms_per_step"])

    if rank == 0 and not args.no_cpu_baseline and world == 1:
        from oracle import cpu_baseline
        p, logw, loglr = final_params
        inv, _ = host.native.niw_expand(p["R"], want_sigma=False)
        out["cpu_baseline"] = cpu_baseline.run_niw(X, D, K, p["mu"].astype(np.float32), inv.reshape(3 * K, -1).astype(np.float32),
                                                   p["logdet"].astype(np.float32), logw.astype(np.float32), loglr.astype(np.float32), N,
                                                   seconds=args.cpu_seconds)
    if rank == 0:
        # the numbers a reader of the driver's record needs beside the headline, inside a block the driver keeps (`config`)
        also = {}
        if "host_master" in out:
            also["host_master_it_per_s"] = out["host_master"]["it_per_s"]
            also["host_master_ratio_to_headline"] = out["host_master"]["ratio_to_headline"]
            also["host_cpu_model"] = out["host_master"]["cpu_model"]
        if "growth" in out:
            gr = out["growth"]
            also["growth"] = {"it_per_s_whole_run": gr["it_per_s_whole_run"], "K_final": gr["K_final"], "K_true": gr["K_true"],
                              "nmi": gr["nmi_final_vs_generator"], "moving_labels_it_per_s": gr.get("moving_labels", {}).get("it_per_s")}
        lg = out.get("legs", {})
        if "shard8_projection" in lg:
            pj = lg["shard8_projection"]
            also["shard8"] = {k: pj.get(k) for k in ("shard_ms_per_step", "shard_ms_per_step_one_collective_form", "assumed_allreduce_ms_per_step",
                                                     "projected_speedup_1_to_8", "speedup_without_collectives")}
        for name in ("overlap_var4", "overlap_var1", "k256", "c2", "c4", "c5_shard"):
            if name in lg:
                also[name + "_ms_per_step"] = lg[name]["ms_per_step"]
        if "inseparable" in lg:
            also["inseparable_sweep_kernel_ms"] = lg["inseparable"]["sweep_kernel_ms"]
            also["inseparable_full_evals_per_tile"] = lg["inseparable"]["roofline"]["full_evals_per_wave_tile"]
        if "c4" in lg:
            also["c4_traffic_frac"] = lg["c4"]["roofline"].get("traffic_frac")
        out["config"]["also_measured"] = also
        emit(out)
    wk.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
