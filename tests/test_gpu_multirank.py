"""N > 1 on the GPU through the PRODUCT's own path: the real `binding.Worker`, the native engine and the collective INSIDE
libdpmmhip.so (`dpmm_comm_init` / `dpmm_comm_init_host` -> `comm_allreduce` in `run_stats`: the Int64 occupancy all-reduce in front
of the bad-cluster reset, the packed-row all-reduce behind the statistics, the device master on the all-reduced rows).

Stands in for create_suff_stats_dict_node_leader / update_suff_stats_posterior! (src/local_clusters_actions.jl:171-254) and
aggregate_suff_stats (src/priors/niw.jl:64-66, multinomial_prior.jl:41-43).

Two transports, one library path:
  * host transport (gloo all_reduce of the library's pinned staging): two ranks SHARE device 0 -- runs on every GPU box;
  * RCCL: one rank per GPU -- needs >= 2 devices, skipped otherwise.
Every child is a fresh process (spawn); the parent never touches the GPU.  A multi-rank chain must equal the one-rank chain:
the random streams are keyed by the global point index, every rank's engine decides on identical all-reduced rows."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CASES = {
    # name: (prior, D, N, true components, iterations, burnout, device master option (-1 auto / 0 / 1))
    "niw64_dev": ("niw", 64, 200000, 6, 60, 8, 1),
    "niw64_host": ("niw", 64, 200000, 6, 60, 8, 0),
    "niw8": ("niw", 8, 60001, 5, 60, 6, -1),           # (N % 2, N % 3, N % 8 all non-zero)
    "mult100": ("mult", 100, 60000, 6, 50, 6, -1),
    "mult200_dev": ("mult", 200, 40000, 5, 50, 6, 1),      # the Dirichlet draws on each rank's device, from the all-reduced rows
    # DPMM_OPT_ONE_COLLECTIVE = 0: the classic per-step pass (occupancy all-reduce -> reset -> statistics -> row all-reduce) stays covered
    "niw64_dev_classic": ("niw", 64, 200000, 6, 60, 8, 1),
    "niw8_classic": ("niw", 8, 60001, 5, 60, 6, -1),
    "mult100_classic": ("mult", 100, 60000, 6, 50, 6, -1),
}


def _data(host, case):
    prior, D, N, K, iters, burnout, dev = CASES[case]
    if prior == "niw":
        x, y, _, _ = host.generate_gaussian_data(N, D, K, 100.0, seed=4242)
        hyper = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    else:
        x, y, _ = host.generate_mnmm_data(N, D, K, 80, seed=4242)[:3]
        hyper = host.multinomial_hyper(np.ones(D, np.float32))
    return np.ascontiguousarray(x, np.float32), np.asarray(y), hyper


# how the N points are cut into `world` contiguous shards: None = the product's even split (N r // world); else explicit boundaries
LAYOUTS = {
    "even2": (2, None),
    "uneven3": (3, None),                         # N % 3 != 0 in the cases below: shards of different sizes
    "empty3": (3, lambda N: [0, N // 2, N, N]),   # the last rank holds NO points (n_local = 0): its rows are zeros, its engine still decides
    "tiny3": (3, lambda N: [0, 7, N - 5, N]),     # shards of 7 and 5 points next to a big one: tiles, items and slabs of almost nothing
    "even8": (8, None),                           # what the driver's 8-GPU run does, eight ranks on ONE device over the host transport
}


def _bounds(layout, N):
    world, f = LAYOUTS[layout]
    return f(N) if f is not None else [(N * r) // world for r in range(world + 1)]


def _run(rank, world, port, out, case, backend, layout="even2"):
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    pkg = load_package()
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    from dpmmsubclusters_jl_amd.host.comm import TorchDistComm
    import torch.distributed as dist
    prior, D, N, K, iters, burnout, dev = CASES[case]
    device = rank if backend == "nccl" else 0
    comm = None
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(device))
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{device}"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = TorchDistComm(device=device)
    x, y, hyper = _data(host, case)
    bnd = _bounds(layout, N) if world > 1 else [0, N]
    lo, hi = bnd[rank], bnd[rank + 1]
    wk = pkg.Worker(hyper.kind, D, hi - lo, first_index=lo, device=device, seed=99)
    wk.upload_points(np.ascontiguousarray(x[:, lo:hi].T))
    if case.endswith("_classic"):
        from dpmmsubclusters_jl_amd import binding
        wk.set_option(binding.OPT_ONE_COLLECTIVE, 0)
    s = host.DPMMSampler(wk, hyper, 10.0, N, 99, burnout=burnout, comm=comm)
    if dev >= 0:
        s.model.set_option(engine.OPT_DEVICE_MASTER, dev)
    s.init_first_clusters(1)
    _, nmi, _, kh = s.run_model(iters, gt=y)
    rows = s.model.get("packed")
    lab, sub = (comm.gather_labels(wk) if comm is not None else wk.get_labels())
    info = wk.comm_info()
    cms = wk.last_comm_ms()
    if rank == 0:
        np.savez(out, labels=lab, sub=sub, K=np.array(kh), nmi=np.array(nmi, float), rows=rows, weights=s.weights,
                 logpost=s.log_posterior(), world=info["world"], allreduces=info["allreduces"], rows_bytes=info["rows_bytes"],
                 counts_bytes=info["counts_bytes"], transport=info["transport"], comm_ms=np.array(cms), one_collective=info["one_collective"])
    wk.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(world, port, out, case, backend, layout="even2"):
    mp.spawn(_run, args=(world, port, out, case, backend, layout), nprocs=world, join=True)


def _compare(a, b, case, transport, world=2):
    prior, D, N, K, iters, burnout, dev = CASES[case]
    stride = 1 + D + (D * (D + 1) // 2 if prior == "niw" else 0)
    assert int(b["world"]) == world and str(b["transport"]) == transport
    one = not case.endswith("_classic")                                      # DPMM_OPT_ONE_COLLECTIVE (default while a packed row has <= 4096 doubles): both priors since round 5
    if one:     # ONE all-reduce per step: 3K rows (2K of the labels as swept + K re-drawn left rows); subset passes after splits add theirs
        assert bool(b["one_collective"]) and int(b["rows_bytes"]) == 3 * int(b["K"][-1]) * stride * 8
        assert iters <= int(b["allreduces"]) < iters + 16
    else:
        assert int(b["rows_bytes"]) == 2 * int(b["K"][-1]) * stride * 8 and int(b["counts_bytes"]) == 2 * int(b["K"][-1]) * 8
        assert int(b["allreduces"]) >= 2 * iters                              # occupancies + rows, every step
    assert np.array_equal(a["K"], b["K"]), (a["K"], b["K"])                   # identical split / merge decisions
    assert a["K"][-1] >= K - 1 and b["nmi"][-1] > 0.9
    flips = int((a["labels"] != b["labels"]).sum())
    sflips = int(((a["sub"] != b["sub"]) & (a["labels"] == b["labels"])).sum())
    print(f"{case}/{transport}: K history equal (final {b['K'][-1]}), label flips {flips}/{N}, sub-label flips {sflips}, "
          f"all-reduce ms (counts, rows) {b['comm_ms']}, all-reduces {int(b['allreduces'])} in {iters} steps")
    # The statistics are Float64 sums over the shards in another association than the one-rank pass: rows and parameters agree to ~1e-13
    # relative, so a draw differs only where a uniform falls within that of a CDF edge -- not once in these chains (measured: 0 flips in
    # every case of rounds 3 and 4).  The test asserts what is measured: the SAME chain.
    assert flips == 0 and sflips == 0, (flips, sflips)
    np.testing.assert_allclose(a["rows"], b["rows"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(a["weights"], b["weights"], rtol=1e-6)
    assert abs(float(a["logpost"]) - float(b["logpost"])) <= 1e-9 * abs(float(a["logpost"]))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", list(CASES))
def test_two_ranks_share_one_gpu_host_transport(tmp_path, case):
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    o1, o2 = str(tmp_path / "r1.npz"), str(tmp_path / "r2.npz")
    port = 29700 + 2 * list(CASES).index(case)
    _spawn(1, port, o1, case, "gloo")
    _spawn(2, port + 1, o2, case, "gloo")
    _compare(np.load(o1), np.load(o2), case, "host")


MORE_RANKS = [("niw8", "uneven3"), ("niw8", "empty3"), ("niw64_dev", "tiny3"), ("mult100", "empty3"), ("niw8", "even8"), ("niw64_dev", "even8"),
              ("mult200_dev", "even8")]


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("case,layout", MORE_RANKS)
def test_more_ranks_and_ragged_shards_share_one_gpu(tmp_path, case, layout):
    """World sizes 3 and 8, N % world != 0, a rank WITHOUT points and shards of a handful of points -- through the library's own collective path
    (host transport: all ranks on device 0).  The chain must be the one-rank chain: K history, every label and sub-label, rows to 1e-11.
    The driver's first real 8-rank run must not be the first 8-rank run of this code."""
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    world = LAYOUTS[layout][0]
    o1, o2 = str(tmp_path / "r1.npz"), str(tmp_path / "rw.npz")
    port = 29800 + 3 * MORE_RANKS.index((case, layout))
    _spawn(1, port, o1, case, "gloo")
    _spawn(world, port + 1, o2, case, "gloo", layout)
    _compare(np.load(o1), np.load(o2), case, "host", world)


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL leg needs one GPU per rank")
@pytest.mark.parametrize("case", list(CASES))
def test_two_ranks_rccl(tmp_path, case):
    o1, o2 = str(tmp_path / "r1.npz"), str(tmp_path / "r2.npz")
    port = 29740 + 2 * list(CASES).index(case)
    _spawn(1, port, o1, case, "nccl")
    _spawn(2, port + 1, o2, case, "nccl")
    _compare(np.load(o1), np.load(o2), case, "rccl")


@pytest.mark.timeout(600)
def test_attach_refuses_to_stay_local(tmp_path):
    """A multi-rank group whose worker cannot be attached must raise, not run on local statistics."""
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from dpmmsubclusters_jl_amd.host.comm import TorchDistComm

    class NoAttach:
        pass
    c = TorchDistComm.__new__(TorchDistComm)
    c.rank, c.world, c.backend = 0, 2, "gloo"
    with pytest.raises(RuntimeError):
        c.attach(NoAttach())


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks_and_reports_the_collective(tmp_path):
    """`bench.py --gpus 2` without a launcher: the script starts two ranks itself (child torchrun before any GPU call), the line says
    n_gpus = 2 and what the collective saw.  On a one-GPU box the two ranks share the device (--share-gpu: gloo + host transport)."""
    import json
    import subprocess
    share = [] if torch.cuda.device_count() >= 2 else ["--share-gpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--points", "400000", "--steps", "5", "--warmup", "1",
                        "--settle", "5", "--blocks", "1", "--no-legs", "--no-cpu-baseline", "--no-dense", "--growth-iters", "40"] + share,
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    short = json.loads(line)                               # the contract line: compact, with the collective's summary
    assert len(line) < 4096 and short["n_gpus"] == 2 and short["comm"]["world"] == 2 and short["comm"]["allreduces_per_step"] == 1.0
    d = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("bench.py details: ")][-1][len("bench.py details: "):])      # everything the run measured
    assert d["value"] == short["value"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["comm"]["world"] == 2 and d["comm"]["transport"] == ("host" if share else "rccl")
    assert d["comm"]["rows_allreduce_bytes"] == 3 * 32 * (1 + 64 + 64 * 65 // 2) * 8 and d["comm"]["rows_allreduce_ms"] > 0
    assert d["comm"]["one_collective_per_step_pass"] and d["comm"]["allreduces_per_step_in_timed_block"] == 1.0
    assert d["config"]["points_per_gpu"] == 200000 and d["growth"]["K_final"] >= 2


@pytest.mark.timeout(1500)
def test_bench_with_eight_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 8 --share-gpu`: the whole 8-rank bench path (own launcher, barrier + max-over-ranks timing, the collective's report,
    growth run, host-master leg) on one device over the host transport, small N."""
    import json
    import subprocess
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--points", "800000", "--steps", "5", "--warmup", "1",
                        "--settle", "5", "--blocks", "1", "--no-legs", "--no-cpu-baseline", "--no-dense", "--growth-iters", "40"],
                       env=env, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-2000:]
    short = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert short["n_gpus"] == 8 and short["comm"]["world"] == 8
    d = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("bench.py details: ")][-1][len("bench.py details: "):])
    assert d["n_gpus"] == 8 and d["value"] > 0 and d["comm"]["world"] == 8 and d["comm"]["transport"] == "host"
    assert d["config"]["points_per_gpu"] == 100000 and d["growth"]["K_final"] >= 2 and d["host_master"]["it_per_s"] > 0


# ---------------------------------------------------------------------------------------------- a dead peer is an error, not a hang
def _dying_rank(rank, world, port, out):
    """Two ranks over the host transport; rank 1 leaves after three sweeps without saying goodbye."""
    sys.path.insert(0, ROOT)
    import datetime
    from __graft_entry__ import load_package
    pkg = load_package()
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    from dpmmsubclusters_jl_amd.host.comm import TorchDistComm
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))
    comm = TorchDistComm(device=0)
    N, D = 40000, 8
    x, y, _, _ = host.generate_gaussian_data(N, D, 4, 100.0, seed=5)
    x = np.ascontiguousarray(x, np.float32)
    lo, hi = (N * rank) // world, (N * (rank + 1)) // world
    wk = pkg.Worker(0, D, hi - lo, first_index=lo, device=0, seed=3)
    wk.upload_points(np.ascontiguousarray(x[:, lo:hi].T))
    s = host.DPMMSampler(wk, host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D)), 10.0, N, 3, burnout=5, comm=comm)
    s.init_first_clusters(1)
    res = "no error"
    try:
        for it in range(12):
            if rank == 1 and it == 3:
                os._exit(0)                      # gone: no barrier, no destroy
            s.group_step(False, False)
    except Exception as e:  # noqa: BLE001
        res = f"{type(e).__name__}: {e}"
    with open(out, "w") as f:
        f.write(res)
    os._exit(0)


@pytest.mark.timeout(300)
def test_dead_peer_becomes_ecomm_not_a_hang(tmp_path):
    """A rank that dies between two sweeps: the survivor's next statistics pass fails with DPMM_ECOMM (-6) within the transport's time-out
    instead of waiting for ever (the reference's master would: SURVEY section 5)."""
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    import time
    out = str(tmp_path / "rank0.txt")
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_dying_rank, args=(r, 2, 29900, out)) for r in range(2)]
    t0 = time.time()
    for p in ps:
        p.start()
    for p in ps:
        p.join(240)
    assert not any(p.is_alive() for p in ps), "a rank is still waiting for its dead peer"
    txt = open(out).read()
    print(f"survivor after {time.time() - t0:.1f} s: {txt[:300]}")
    # (through the native engine the worker's DPMM_ECOMM arrives as the engine's error carrying the worker's message)
    assert ("Error" in txt) and ("all-reduce callback failed" in txt or "error -6" in txt), txt


def _aborting_rank(out):
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    pkg = load_package()
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    from dpmmsubclusters_jl_amd import binding
    N, D = 2000000, 64
    X, y = host.gaussian_mixture_shard(N, D, 32, 100.0, 12345, 0, N)
    wk = pkg.Worker(0, D, N, device=0, seed=3)
    wk.upload_points(X)
    wk.set_option(binding.OPT_SCREEN_MARGIN, 0)                  # every cluster evaluated in full: a step of ~5 ms, i.e. host waits well past 1 ms
    wk.comm_init(wk.comm_unique_id(), 0, 1)                      # a one-rank RCCL communicator: the library's collective path and its watchdog
    s = host.DPMMSampler(wk, host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D)), 10.0, N, 3, burnout=5)
    s.start_from_labels(y, 1 + (np.arange(N) & 1), 32)
    res = []
    try:
        for _ in range(3):
            s.group_step(False, False)
        res.append("three steps fine")
        wk.set_option(binding.OPT_COMM_TIMEOUT_MS, 1)            # every wait of a step is now "too long": the watchdog's next look (every 20 ms) aborts
        for _ in range(4000):
            s.group_step(False, False)
        res.append("no error")
    except Exception as e:  # noqa: BLE001
        res.append(f"{type(e).__name__}: {e}")
    with open(out, "w") as f:
        f.write(" | ".join(res))
    os._exit(0)


@pytest.mark.timeout(300)
def test_rccl_watchdog_aborts_a_wait_past_its_deadline(tmp_path):
    """DPMM_OPT_COMM_TIMEOUT_MS on the RCCL transport: a host call that blocks on the ctx stream longer than the limit gets its communicator
    aborted by the watchdog and returns DPMM_ECOMM.  With one GPU there is no peer to kill, so the limit is set absurdly low (1 ms) on a
    one-rank communicator instead: the same watchdog, the same ncclCommAbort, the same error path."""
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    out = str(tmp_path / "res.txt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_aborting_rank, args=(out,))
    p.start()
    p.join(240)
    assert not p.is_alive(), "the process hangs"
    txt = open(out).read()
    print(txt[:300])
    assert txt.startswith("three steps fine") and "Error" in txt and "timed out" in txt, txt
