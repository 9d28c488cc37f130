"""Checkpoint / resume, the parameter-file mode and .npy ingestion -- host logic on the CPU with the
oracle-backed FakeWorker (test-only).  The GPU counterparts are in test_gpu_checkpoint.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    from __graft_entry__ import load_package
    load_package()
    import importlib
    return importlib.import_module("dpmmsubclusters_jl_amd.host")


def _data(host, n=600, D=2, K=3, seed=5):
    x, y = host.generate_gaussian_data(n, D, K, 60.0, seed=seed)[:2]
    return x.astype(np.float32), y


def test_load_data_matches_reference_semantics(host, tmp_path):
    # utils.jl:5-14: Samples x Dimensions on disk, NaN -> 0, transposed to Dimensions x Samples
    a = np.arange(12, dtype=np.float64).reshape(4, 3)
    a[1, 2] = np.nan; a[3, 0] = np.nan
    np.save(tmp_path / "bob.npy", a)
    got = host.load_data(str(tmp_path) + "/", prefix="bob")
    exp = np.nan_to_num(a, nan=0.0).T
    assert got.shape == (3, 4) and np.array_equal(got, exp)
    raw = host.load_data(str(tmp_path) + "/", prefix="bob", swapDimension=False)
    assert raw.shape == (4, 3) and np.array_equal(raw, exp.T)
    mm = host.checkpoint.load_data(str(tmp_path) + "/", "bob", swapDimension=False, mmap=True)
    assert isinstance(mm, np.memmap) and np.isnan(mm[1, 2])      # untouched: cleaned on the device


PARAMS = """
data_path = {path!r}
data_prefix = "pts"
iterations = {iters}
hard_clustering = false
initial_clusters = 2
argmax_sample_stop = 1
split_stop = 1
random_seed = 77
burnout_period = 3
max_clusters = Inf
α = 10.0
hyper_params = DPMMSubClusters.niw_hyperparams(1.0, zeros(2), 5, eye(2))
enable_saving = true
model_save_interval = 4
save_path = {save!r}
save_file_prefix = "checkpoint_"
"""


def test_read_params_defaults_and_overrides(host, tmp_path):
    f = tmp_path / "params.py"
    f.write_text(PARAMS.format(path=str(tmp_path) + "/", iters=9, save=str(tmp_path) + "/"))
    P = host.checkpoint.read_params(str(f))
    assert P["iterations"] == 9 and P["alpha"] == 10.0 and P["burnout_period"] == 3 and P["model_save_interval"] == 4
    assert P["max_split_iter"] == 20 and P["smart_splits"] is False and P["overwrite_prec"] is False   # global_params.jl defaults
    assert P["hyper_params"].dim == 2 and P["hyper_params"].nu == 5.0


def test_basic_mode_checkpoint_resume_continues_same_chain(host, tmp_path):
    from fake_worker import FakeWorker
    x, y = _data(host)
    kw = dict(seed=11, burnout=3, verbose=False, worker_factory=FakeWorker, nthreads=1)
    hyper = host.niw_hyperparams(1.0, np.zeros(2), 5, np.eye(2))
    full, *_ = host.dp_parallel(x, hyper, 10.0, 8, 2, save_model=True, save_path=str(tmp_path) + "/", model_save_interval=3, **kw)
    assert [os.path.basename(f) for f in full.checkpoints] == ["checkpoint__3.npz", "checkpoint__6.npz"]   # path*prefix*"_"*iter
    ck = host.load_checkpoint(full.checkpoints[0])
    assert int(ck["iter"]) == 3 and ck["labels"].shape == (x.shape[1],) and str(ck["global_params"]) == "none"
    res, it, *_ = host.resume_from_checkpoint(full.checkpoints[0], x, 8, verbose=False, worker_factory=FakeWorker, nthreads=1)
    assert len(it) == 5                                            # iterations 4..8
    assert np.array_equal(res.labels, full.labels) and np.array_equal(res.labels_subcluster, full.labels_subcluster)
    assert res.sampler.K == full.sampler.K and np.array_equal(res.sampler.weights, full.sampler.weights)
    with pytest.raises(ValueError):
        host.run_model_from_checkpoint(full.checkpoints[0], worker_factory=FakeWorker)   # basic-mode file: no parameter file


def test_advanced_mode_and_run_model_from_checkpoint(host, tmp_path):
    from fake_worker import FakeWorker
    x, y = _data(host, seed=9)
    rows = x.T.astype(np.float64).copy()
    rows[5, 1] = np.nan                                            # cleaned at ingestion
    np.save(tmp_path / "pts.npy", rows)
    f = tmp_path / "params.py"
    f.write_text(PARAMS.format(path=str(tmp_path) + "/", iters=9, save=str(tmp_path) + "/ck/"))
    full, it, nmi, lik, kh = host.dp_parallel(str(f), verbose=False, gt=y, worker_factory=FakeWorker, nthreads=1)
    assert len(it) == 9 and len(kh) == 9 and isinstance(nmi[-1], float)
    assert [os.path.basename(c) for c in full.checkpoints] == ["checkpoint__4.npz", "checkpoint__8.npz"]
    clean = np.nan_to_num(rows, nan=0.0).astype(np.float32)
    assert np.array_equal(full.sampler.wk.X, clean)
    res, it2, *_ = host.run_model_from_checkpoint(full.checkpoints[0], verbose=False, worker_factory=FakeWorker, nthreads=1)
    assert len(it2) == 5
    assert np.array_equal(res.labels, full.labels) and res.sampler.K == full.sampler.K
    assert np.allclose(res.sampler.N, full.sampler.N)


def test_multinomial_resume_continues_same_chain(host, tmp_path):
    # the Dirichlet draws use pre-generated first-trial noise in a running chain and generate the same stream inline in the first
    # sweep after a resume: both must give the same parameters, hence the same labels
    from fake_worker import FakeWorker
    rng = np.random.default_rng(3)
    P = rng.dirichlet(np.ones(12) * 0.3, size=3)
    y = rng.integers(0, 3, 500)
    x = np.stack([rng.multinomial(40, P[k]) for k in y]).T.astype(np.float32)      # D x N
    hyper = host.multinomial_hyper(np.ones(12, np.float32))
    kw = dict(seed=5, burnout=3, verbose=False, worker_factory=FakeWorker, nthreads=2)
    full, *_ = host.dp_parallel(x, hyper, 10.0, 9, 1, save_model=True, save_path=str(tmp_path) + "/", model_save_interval=4, **kw)
    res, it, *_ = host.resume_from_checkpoint(full.checkpoints[0], x, 9, verbose=False, worker_factory=FakeWorker, nthreads=2)
    assert len(it) == 5
    assert np.array_equal(res.labels, full.labels) and np.array_equal(res.labels_subcluster, full.labels_subcluster)
    assert res.sampler.K == full.sampler.K and np.array_equal(res.sampler.weights, full.sampler.weights)
