"""GPU: .npy ingestion on the device, checkpoint / resume through the C ABI."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_checkpoint_cpu import PARAMS  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="module")
def host(pkg):
    import importlib
    return importlib.import_module("dpmmsubclusters_jl_amd.host")


@pytest.mark.parametrize("dtype,D,n", [(np.float64, 5, 3001), (np.float32, 64, 2000), (np.float64, 20, 513)])
def test_npy_ingestion_on_device(pkg, dtype, D, n):
    rng = np.random.default_rng(3)
    rows = rng.normal(size=(n, D)).astype(dtype) * 3
    rows[rng.integers(0, n, 40), rng.integers(0, D, 40)] = np.nan          # utils.jl:9-13: NaN -> 0
    clean = np.nan_to_num(rows, nan=0.0).astype(np.float32)
    lab = rng.integers(1, 4, n); sub = rng.integers(1, 3, n)
    out = []
    for mode in ("npy", "plain"):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1)
        if mode == "npy":
            wk.upload_points_npy(rows)
        else:
            wk.upload_points(clean)
        wk.set_labels(lab, sub); wk.set_num_clusters(3)
        N, sums, S = wk.unpack(wk.suffstats_packed(None), 3)
        out.append((N.copy(), sums.copy(), S.copy()))
        wk.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)                                       # same device contents, bit for bit
    N, sums, S = out[0]
    assert not np.isnan(sums).any()
    for k in range(3):
        m = lab == k + 1
        assert N[k, 0] == m.sum()
        assert np.allclose(sums[k, 0], clean[m].astype(np.float64).sum(0), rtol=1e-12, atol=1e-9)


def test_advanced_mode_checkpoint_and_resume_on_gpu(host, tmp_path):
    x, y = host.generate_gaussian_data(4000, 2, 4, 80.0, seed=21)[:2]
    rows = x.T.astype(np.float64).copy()
    rows[17, 0] = np.nan
    np.save(tmp_path / "pts.npy", rows)
    f = tmp_path / "params.py"
    f.write_text(PARAMS.format(path=str(tmp_path) + "/", iters=12, save=str(tmp_path) + "/ck/"))
    full, it, nmi, lik, kh = host.dp_parallel(str(f), verbose=False, gt=y)
    assert len(it) == 12 and [os.path.basename(c) for c in full.checkpoints] == ["checkpoint__4.npz", "checkpoint__8.npz", "checkpoint__12.npz"]
    ck = host.load_checkpoint(full.checkpoints[1])
    assert int(ck["iter"]) == 8 and ck["labels"].min() >= 1 and set(np.unique(ck["labels_subcluster"])) <= {1, 2}
    res, it2, *_ = host.run_model_from_checkpoint(full.checkpoints[1], verbose=False)
    assert len(it2) == 4
    assert np.array_equal(res.labels, full.labels) and np.array_equal(res.labels_subcluster, full.labels_subcluster)
    assert res.sampler.K == full.sampler.K and np.array_equal(res.sampler.weights, full.sampler.weights)


def test_fit_save_model_and_resume_on_gpu(host, tmp_path):
    x, y = host.generate_gaussian_data(3000, 3, 3, 60.0, seed=4)[:2]
    r = host.fit(x.astype(np.float32), 10.0, iters=10, seed=5, burnout=4, verbose=False, save_model=True,
                 save_path=str(tmp_path) + "/", model_save_interval=5)
    model = r[8]
    assert len(model.checkpoints) == 2
    res, *_ = host.resume_from_checkpoint(model.checkpoints[0], x.astype(np.float32), 10, verbose=False)
    assert np.array_equal(res.labels, r[0]) and res.sampler.K == model.sampler.K


def test_resume_continues_the_chain_with_the_device_master(host, tmp_path):
    """D >= 128: posteriors, factorisations and draws run on the device (csrc/niw_master.hip).  A resumed run uploads the saved
    statistics rows, so the device produces the same posteriors and -- with the restored epoch counters -- the same draws."""
    x, y = host.generate_gaussian_data(4000, 130, 3, 60.0, seed=8)[:2]
    x = x.astype(np.float32)
    r = host.fit(x, 10.0, iters=10, seed=5, burnout=4, verbose=False, save_model=True, save_path=str(tmp_path) + "/", model_save_interval=5)
    model = r[8]
    assert len(model.checkpoints) == 2
    res, *_ = host.resume_from_checkpoint(model.checkpoints[0], x, 10, verbose=False)
    assert np.array_equal(res.labels, r[0]) and np.array_equal(res.labels_subcluster, model.labels_subcluster)
    assert res.sampler.K == model.sampler.K and np.array_equal(res.sampler.weights, model.sampler.weights)
