"""Outlier component (second half) -- host logic with the oracle-backed FakeWorker (test-only).
Reference: create_outlier_local_cluster (local_clusters_actions.jl:42-61), sample_clusters! (:417-437),
check_and_split! (:348-350), init (dp-parallel-sampling.jl:49, 63-65)."""
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


def outlier_problem(n=1500, frac=0.08, seed=3):
    rng = np.random.default_rng(seed)
    cent = np.array([[-10.0, 0.0], [10.0, 0.0], [0.0, 12.0]])
    y = rng.integers(0, 3, n)
    x = cent[y] + rng.normal(size=(n, 2)) * 0.8
    out = rng.random(n) < frac
    x[out] = rng.uniform(-40, 40, size=(out.sum(), 2))
    return x.T.astype(np.float32), y, out


def run_outlier_fit(host, worker_factory=None, **kw):
    x, y, out = outlier_problem()
    hyper = host.niw_hyperparams(1.0, np.zeros(2), 5, np.eye(2))
    ohyper = host.niw_hyperparams(1.0, np.zeros(2), 5, np.eye(2) * 400.0)
    r = host.fit(x, hyper, 10.0, iters=60, seed=21, burnout=5, verbose=False, outlier_weight=0.05, outlier_params=ohyper,
                 worker_factory=worker_factory, **kw)
    return r, out


def check_outlier_result(r, out):
    """The reference's outlier component (outlier_mod > 0): cluster 1 has its OWN prior (outlier_hyper_params), a constant
    weight, is never split / merged / removed, and is otherwise an ordinary cluster -- its statistics and posterior follow the
    points labelled 1 (update_suff_stats_posterior!, local_clusters_actions.jl:237-251) and its distribution is re-drawn every
    sweep (sample_cluster_params mutates it in place; only the fetch is skipped, :424-427)."""
    labels, clusters, weights, model = r[0], r[1], r[2], r[8]
    s = model.sampler
    assert weights[0] == np.float32(0.05)                           # constant weight, first component
    assert abs(float(weights[1:].sum()) - 0.95) < 0.95 * 0.2       # the rest is Dirichlet mass times (1 - outlier_mod), minus the alpha share
    N, sums, S, post = s.N, s.sums, s.S, s.post
    assert s.points_count[0] == int(N[0, 0]) == int((labels == 1).sum())
    # rows 0..2 (the outlier's cluster / left / right) carry posteriors under the OUTLIER prior, every other row under the cluster prior
    want = s.outlier_prior.posterior(N[0], sums[0], S[0])
    for key in ("kappa", "nu", "m", "logdet_psi"):
        np.testing.assert_allclose(post[key][0:3], want[key], rtol=1e-12, atol=1e-12)
    want1 = s.prior.posterior(N[1], sums[1], S[1])
    np.testing.assert_allclose(post["logdet_psi"][3:6], want1["logdet_psi"], rtol=1e-12, atol=1e-12)
    assert labels.min() >= 1
    is_out = labels == 1
    # the broad component only ever takes background points (the DP is free to open broad clusters of its own too)
    assert is_out[out].mean() > 0.05 and is_out[~out].mean() < 0.02, (is_out[out].mean(), is_out[~out].mean())
    assert len(np.unique(labels[~out])) >= 3


def test_outlier_component_with_fake_worker(host):
    from fake_worker import FakeWorker
    r, out = run_outlier_fit(host, FakeWorker, nthreads=1)
    check_outlier_result(r, out)


def test_outlier_argument_checks(host):
    from fake_worker import FakeWorker
    x, y, out = outlier_problem(300)
    with pytest.raises(TypeError):
        host.fit(x, 10.0, iters=2, outlier_weight=0.05, worker_factory=FakeWorker, verbose=False)
    with pytest.raises(ValueError):
        host.fit(x, 10.0, iters=2, outlier_weight=0.05, outlier_params=host.multinomial_hyper(np.ones(2)), worker_factory=FakeWorker, verbose=False)
