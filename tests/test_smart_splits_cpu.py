"""Smart splits -- host logic with the oracle-backed FakeWorker (test-only)."""
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


def two_blob_problem(n=4000, D=3, seed=2):
    rng = np.random.default_rng(seed)
    z = rng.integers(0, 2, n)
    X = (rng.normal(size=(n, D)) + np.where(z[:, None] == 1, 6.0, -6.0)).astype(np.float32)
    return X, z


def reference_smart_init(X, labels, k1, max_iter=20):
    """Plain restatement of smart_cluster_init! (local_clusters_actions.jl:555-627) for ONE worker, Float64."""
    m = labels == k1
    pts = X[m].astype(np.float64)
    N = pts.shape[0]
    mu = pts.sum(0) / N
    M = (pts.T @ pts) / N - np.outer(mu, mu)
    M = 0.5 * (M + M.T)
    vals, vecs = np.linalg.eigh(M)
    v1 = vecs[int(np.argmax(vals)), :]
    t = (pts - mu) @ v1
    lo, hi = np.quantile(t, 0.001), np.quantile(t, 0.009)
    it, conv = 0, False
    while it < max_iter and not conv:
        s1 = np.abs(t - lo) < np.abs(t - hi)
        with np.errstate(invalid="ignore", divide="ignore"):
            nlo, nhi = t[s1].sum() / s1.sum(), t[~s1].sum() / (~s1).sum()
        if nlo == lo and nhi == hi:
            conv = True
        else:
            lo, hi = nlo, nhi
        it += 1
    return np.where(np.abs(t - lo) < np.abs(t - hi), 1, 2), (lo, hi, it)


def make_sampler(host, worker_cls, X, **kw):
    n, D = X.shape
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    wk = worker_cls(0, D, n, **kw)
    wk.upload_points(X)
    s = host.DPMMSampler(wk, prior, 10.0, n, 3, burnout=5, nthreads=1)
    return s


def test_smart_cluster_init_follows_the_reference_steps(host):
    from fake_worker import FakeWorker
    X, z = two_blob_problem()
    s = make_sampler(host, FakeWorker, X, first_index=0, device=0, seed=3)
    s.start_from_labels(np.ones(len(X), np.int64), 1 + np.random.default_rng(0).integers(0, 2, len(X)), 1)
    s.smart_cluster_init(0)
    _, sub = s.wk.get_labels()
    exp, (lo, hi, it) = reference_smart_init(X, np.ones(len(X), np.int64), 1)
    assert it <= 20 and np.isfinite(lo) and np.isfinite(hi)
    agree = (sub == exp).mean()
    assert agree == 1.0 or agree == 0.0 or agree > 0.999, agree     # S comes from packed Float64 sums: same partition
    assert set(np.unique(sub)) <= {1, 2}


def test_fit_with_smart_splits_recovers_clusters(host):
    from fake_worker import FakeWorker
    rng = np.random.default_rng(4)
    cent = np.array([[-12.0, 0.0], [12.0, 0.0], [0.0, 14.0]])
    y = rng.integers(0, 3, 1200)
    x = (cent[y] + rng.normal(size=(1200, 2))).T.astype(np.float32)
    r = host.fit(x, 10.0, iters=50, seed=12, burnout=5, verbose=False, gt=y, smart_splits=True, worker_factory=FakeWorker, nthreads=1)
    assert len(np.unique(r[0])) == 3 and r[4][-1] > 0.95, (r[6][-1], r[4][-1])
    with pytest.raises(ValueError):
        host.fit(np.abs(x).astype(np.float32), host.multinomial_hyper(np.ones(2)), 10.0, iters=2, smart_splits=True,
                 worker_factory=FakeWorker, verbose=False)
