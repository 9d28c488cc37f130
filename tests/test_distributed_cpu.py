"""N>1 path on CPU: two `gloo` ranks run the real host sampler + comm layer (all-reduce of the packed
sufficient statistics) over a test-only oracle-backed worker and must reproduce the single-rank run."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    rng = np.random.default_rng(5)
    means = np.array([[-6.0, 0.0], [6.0, 0.0], [0.0, 7.0]])
    z = rng.integers(0, 3, 1500)
    x = (means[z] + rng.normal(size=(1500, 2)) * 0.7).T.astype(np.float32)
    return x, z + 1


def _run(rank, world, port, out, smart=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package
    load_package()
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    from fake_worker import FakeWorker
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from dpmmsubclusters_jl_amd.host.comm import TorchDistComm
        comm = TorchDistComm()
    x, y = _problem()
    res = host.fit(x, 10.0, iters=80, seed=31, burnout=5, verbose=False, gt=y, comm=comm, worker_factory=FakeWorker, nthreads=1,
                   smart_splits=smart)
    if rank == 0:
        np.savez(out, labels=res[0], K=np.array(res[6]), nmi=np.array(res[4], float), weights=res[2], sub=res[7])
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def _spawn(world, port, out, smart=False):
    if world == 1:
        _run(0, 1, port, out, smart)
    else:
        mp.spawn(_run, args=(world, port, out, smart), nprocs=world, join=True)


@pytest.mark.timeout(600)
def test_two_ranks_match_one_rank(tmp_path):
    o1, o2 = str(tmp_path / "r1.npz"), str(tmp_path / "r2.npz")
    _spawn(1, 29611, o1)
    _spawn(2, 29612, o2)
    a, b = np.load(o1), np.load(o2)
    assert a["K"][-1] == 3 and b["K"][-1] == 3
    assert np.array_equal(a["K"], b["K"])                       # identical split/merge decisions on every rank layout
    assert (a["labels"] != b["labels"]).mean() < 1e-3           # index-keyed RNG: sharding does not change the draws
    assert a["nmi"][-1] > 0.95 and b["nmi"][-1] > 0.95
    np.testing.assert_allclose(a["weights"], b["weights"], rtol=1e-5)


@pytest.mark.timeout(600)
def test_smart_splits_two_ranks_match_one_rank(tmp_path):
    """smart_cluster_init! across shards: every rank derives the SAME direction and centre from the all-reduced statistics,
    the percentile seeds and 2-means sums are reduced over the ranks -- the chain must equal the single-rank chain."""
    o1, o2 = str(tmp_path / "s1.npz"), str(tmp_path / "s2.npz")
    _spawn(1, 29615, o1, True)
    _spawn(2, 29616, o2, True)
    a, b = np.load(o1), np.load(o2)
    assert np.array_equal(a["K"], b["K"]) and a["K"][-1] == 3
    assert (a["labels"] != b["labels"]).mean() < 1e-3 and (a["sub"] != b["sub"]).mean() < 5e-3
    assert b["nmi"][-1] > 0.95


def test_fake_worker_matches_packed_contract():
    """The fake worker's packed rows follow include/dpmm_hip.h (so the CPU test exercises the real unpack path shape)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_worker import FakeWorker
    rng = np.random.default_rng(0)
    X = rng.normal(size=(200, 3)).astype(np.float32)
    w = FakeWorker(0, 3, 200)
    w.upload_points(X)
    w.set_labels(rng.integers(1, 4, 200), rng.integers(1, 3, 200))
    w.set_num_clusters(3)
    pk = w.suffstats_packed()
    assert pk.shape == (6, 1 + 3 + 6)
    N, s, S = w.unpack(pk)
    from oracle import oracle as orc
    oN, os_, oS = orc.suffstats_niw(X, 3, w.labels, w.sub, 3)
    np.testing.assert_allclose(N, oN); np.testing.assert_allclose(S, oS, rtol=1e-12, atol=1e-12)
