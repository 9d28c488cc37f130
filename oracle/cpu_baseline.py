"""CPU baseline of the benchmark: the reference ALGORITHM's worker path on the host cores, structured like the reference
runs it -- P worker PROCESSES, each holding a contiguous column range of the points and one BLAS thread
(`addprocs(P)` + `BLAS.set_num_threads(1)`: /root/reference README.md:43, docs/src/perf.md:6-8), timed like the reference
times an iteration (src/dp-parallel-sampling.jl:363-366): wall clock from the moment every worker starts its sweep to the
moment the last one has returned its sufficient statistics, plus the master's serial steps.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg); never imported by the product.

Each worker runs `oracle.sweep_numpy_niw` -- per-cluster GEMM invSigma*z + column dot (mv_gaussian.jl:21-25), the n x K
table, max-shift/exp/normalise + inverse-CDF scan (utils.jl:19-31), gathered views per cluster for the sub-labels
(local_clusters_actions.jl:77-78) and three Float64 statistic passes per cluster (:158-166, niw.jl:42-51) -- on its range, in
chunks of CHUNK points (bounds the D x n temporaries; the arithmetic is per point, so chunking changes nothing else).
The master part times calc_posterior + log_marginal_likelihood for the 3K statistic sets and the K(K-1)/2 pooled pairs of
check_and_merge! in numpy (serial, as the reference's master is).

`julia` is probed first: if a Julia with DPMMSubClusters is on the box the genuine reference
would be timed instead; this image has none, so kind is "port" (`parallelism`: "multiprocess").
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

CHUNK = 100000


def cpu_budget():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _worker_main(path, lo, hi, start_at):
    """One worker process: its column range [lo, hi) of the sample, one BLAS thread."""
    from threadpoolctl import threadpool_limits
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as orc
    z = np.load(os.path.join(path, "params.npz"))
    D, K = int(z["D"]), int(z["K"])
    X = np.load(os.path.join(path, "X.npy"), mmap_mode="r")[lo:hi]
    X = np.ascontiguousarray(X)
    n = X.shape[0]
    u0, u1 = orc.uniforms(1, 1, 0, lo, n)
    args = (z["mu"], z["invS"], z["logdet"], z["logw"], z["loglr"])
    with threadpool_limits(limits=1, user_api="blas"):
        orc.sweep_numpy_niw(X[:2000], D, *args, u0[:2000], u1[:2000])      # warm-up (page in BLAS, allocate)
        while time.time() < start_at:
            time.sleep(0.001)
        t0 = time.time()
        Ns = np.zeros((K, 3)); sums = np.zeros((K, 3, D)); Ss = np.zeros((K, 3, D, D))
        for a in range(0, n, CHUNK):
            b = min(n, a + CHUNK)
            _, _, (N1, s1, S1) = orc.sweep_numpy_niw(X[a:b], D, *args, u0[a:b], u1[a:b])
            Ns += N1; sums += s1; Ss += S1
        t1 = time.time()
    print(json.dumps({"lo": lo, "hi": hi, "t0": t0, "t1": t1, "N": float(Ns[:, 0].sum())}))


def _master_seconds(D, K, rng):
    """The master's serial share of an iteration in the reference: 3K posteriors + marginals, K(K-1)/2 pooled pairs."""
    from oracle import oracle as orc
    prior = (1.0, np.zeros(D), D + 3.0, np.eye(D))
    A = rng.normal(size=(D + 8, D))
    S = A.T @ A * 1000.0
    sm = rng.normal(size=D) * 100.0
    t0 = time.perf_counter()
    for _ in range(3 * K):
        post = orc.niw_calc_posterior(*prior, 3000.0, sm, S)
        orc.niw_log_marginal(prior, post, 3000.0, D)
    for _ in range(K * (K - 1) // 2):
        post = orc.niw_calc_posterior(*prior, 6000.0, 2 * sm, 2 * S)
        orc.niw_log_marginal(prior, post, 6000.0, D)
    return time.perf_counter() - t0


def run_niw(X, D, K, mu, invS, logdet, logw, loglr, N_total, seconds=20.0, procs=None):
    """Time one sweep of the reference algorithm's worker path with P processes on a bounded sample of X (rows = points);
    returns the `cpu_baseline` object of the bench line."""
    julia = shutil.which("julia")
    P = int(procs or cpu_budget())
    from threadpoolctl import threadpool_limits
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as orc
    # calibrate: points per second of ONE worker (in-process, one BLAS thread)
    ncal = min(20000, X.shape[0])
    u0, u1 = orc.uniforms(1, 1, 0, 0, ncal)
    with threadpool_limits(limits=1, user_api="blas"):
        orc.sweep_numpy_niw(X[:2000], D, mu, invS, logdet, logw, loglr, u0[:2000], u1[:2000])
        t0 = time.perf_counter()
        orc.sweep_numpy_niw(X[:ncal], D, mu, invS, logdet, logw, loglr, u0, u1)
        rate = ncal / (time.perf_counter() - t0)
    per_proc = int(min(X.shape[0] // P, max(20000, rate * seconds * 0.8)))   # 0.8: P busy processes share caches / memory bandwidth
    m = per_proc * P
    idx = np.linspace(0, X.shape[0] - 1, m).astype(np.int64)      # spans all components
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="dpmm_cpu_baseline_", dir=shm)
    try:
        np.save(os.path.join(tmp, "X.npy"), np.ascontiguousarray(X[idx], np.float32))
        np.savez(os.path.join(tmp, "params.npz"), D=D, K=K, mu=mu, invS=invS, logdet=logdet, logw=logw, loglr=loglr)
        start_at = time.time() + 6.0 + 0.2 * P            # every worker has loaded its range and warmed up by then
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs_ = [subprocess.Popen([sys.executable, "-c",
                                    f"import sys; sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); "
                                    f"from oracle.cpu_baseline import _worker_main; _worker_main({tmp!r}, {r * per_proc}, {(r + 1) * per_proc}, {start_at!r})"],
                                   stdout=subprocess.PIPE, env=env) for r in range(P)]
        outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs_]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    late = max(0.0, max(o["t0"] for o in outs) - start_at)
    wall = max(o["t1"] for o in outs) - start_at
    master = _master_seconds(D, K, np.random.default_rng(0))
    per_iter = wall * (N_total / m) + master
    return {"value": 1.0 / per_iter, "unit": "iterations/s", "cores": P, "kind": "port", "parallelism": "multiprocess",
            "cpu_model": cpu_model(), "julia_found": bool(julia),
            "sample": f"{P} worker processes x {per_proc} points (= {m} of the {N_total} points, {100.0 * m / N_total:.1f} %), one BLAS thread "
                      f"each, K={K}: {wall:.2f} s wall for the sample (latest start +{late:.2f} s), scaled linearly to N, plus {master:.3f} s of "
                      f"serial master maths (3K posteriors + marginals, K(K-1)/2 merge pairs); reference-algorithm restatement "
                      f"(numpy/BLAS), not the Julia package",
            "sample_fraction": m / N_total, "wall_s_sample": wall, "master_s": master}
