"""GPU: worker halves of smart_cluster_init! through the C ABI vs the numpy restatement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_smart_splits_cpu import two_blob_problem, reference_smart_init, make_sampler  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="module")
def host(pkg):
    import importlib
    return importlib.import_module("dpmmsubclusters_jl_amd.host")


@pytest.mark.parametrize("D,n", [(3, 4000), (64, 20011), (20, 777)])
def test_worker_halves_match_numpy(pkg, D, n):
    from fake_worker import FakeWorker
    rng = np.random.default_rng(D)
    X = (rng.normal(size=(n, D)) * 2 + rng.integers(0, 2, n)[:, None] * 5).astype(np.float32)
    lab = rng.integers(1, 4, n); sub = rng.integers(1, 3, n)
    v = rng.normal(size=D); v /= np.linalg.norm(v)
    mu = X[lab == 2].astype(np.float64).mean(0)
    gw = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1); gw.upload_points(X); gw.set_labels(lab, sub); gw.set_num_clusters(3)
    fw = FakeWorker(0, D, n); fw.upload_points(X); fw.set_labels(lab, sub); fw.set_num_clusters(3)
    tg, tf = np.sort(gw.smart_project(2, v, mu)), np.sort(fw.smart_project(2, v, mu))
    assert tg.shape == tf.shape == ((lab == 2).sum(),)
    assert np.allclose(tg, tf, rtol=1e-12, atol=1e-12)            # Float64 dot products, different summation order
    lo, hi = np.quantile(tf, 0.3), np.quantile(tf, 0.7)
    sg, sf = gw.smart_kmeans_iter(2, lo, hi), fw.smart_kmeans_iter(2, lo, hi)
    assert sg[1] == sf[1] and sg[3] == sf[3] and sg[1] + sg[3] == (lab == 2).sum()      # counts exact
    assert np.allclose(sg, sf, rtol=1e-11)
    assert np.array_equal(gw.smart_kmeans_iter(2, lo, hi), sg)    # fixed reduction order: bitwise reproducible
    gw.smart_assign(2, lo, hi); fw.smart_assign(2, lo, hi)
    lg, subg = gw.get_labels(); lf, subf = fw.get_labels()
    assert np.array_equal(lg, lab) and np.array_equal(subg[lab != 2], sub[lab != 2])      # other clusters untouched
    assert (subg != subf).sum() <= 1                                                     # a tie at most
    # empty cluster / errors
    assert len(gw.smart_project(3 + 1, v, mu)) == 0
    with pytest.raises(RuntimeError):
        gw.smart_project(0, v, mu)
    gw.close()


def test_smart_cluster_init_on_gpu_equals_restatement(pkg, host):
    X, z = two_blob_problem(n=6000, D=3, seed=7)
    s = make_sampler(host, pkg.Worker, X, first_index=0, device=0, seed=3)
    s.start_from_labels(np.ones(len(X), np.int64), 1 + np.random.default_rng(0).integers(0, 2, len(X)), 1)
    s.smart_cluster_init(0)
    _, sub = s.wk.get_labels()
    exp, _ = reference_smart_init(X, np.ones(len(X), np.int64), 1)
    agree = (sub == exp).mean()
    assert agree > 0.999 or agree < 0.001, agree


def test_fit_with_smart_splits_on_gpu(host):
    x, y = host.generate_gaussian_data(20000, 2, 6, 100.0, seed=12345)[:2]
    r = host.fit(x.astype(np.float32), 10.0, iters=100, seed=123456789, burnout=10, verbose=False, gt=y, smart_splits=True)
    assert r[4][-1] > 0.9 and 5 <= len(np.unique(r[0])) <= 8, (r[4][-1], len(np.unique(r[0])))
