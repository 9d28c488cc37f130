"""End-to-end `fit` on the GPU, mirroring the reference's own outcome tests
(test/module_tests.jl) and its docs example (docs/src/getting_started.md:27-37 == BASELINE config 1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    from __graft_entry__ import load_package
    load_package()
    import importlib
    return importlib.import_module("dpmmsubclusters_jl_amd.host")


def test_module_deterministic_point_masses(host):
    """test/module_tests.jl:1-32: four point masses, 250 copies each."""
    data = np.zeros((2, 1000), np.float32)
    data[:, 0:250] = [[-1], [-1]]; data[:, 250:500] = [[-1], [1]]
    data[:, 500:750] = [[1], [-1]]; data[:, 750:1000] = [[1], [1]]
    res = host.fit(data, 100.0, iters=200, seed=123456789, burnout=15, verbose=False)
    labels, clusters, weights = res[0], res[1], res[2]
    assert len(clusters) == 4                                   # :21
    assert np.all(weights >= 0.15)                              # :23
    for _, v in host.get_labels_histogram(labels):              # :29-31
        assert v == 250
    assert len(res) == 9 and len(res[3]) == 200 and res[6][-1] == 4
    for c in clusters:                                          # cluster means sit on the masses
        assert np.min(np.abs(np.abs(c.mu) - 1)) < 0.2


def test_docs_example_config1(host):
    """docs/src/getting_started.md:27-37: N=10^4, D=2, 6 components, alpha=10, burnout=10, 100 iterations."""
    x, y, _, _ = host.generate_gaussian_data(10 ** 4, 2, 6, 100.0, seed=4)
    res = host.fit(x, 10.0, iters=100, burnout=10, gt=y, seed=12345, verbose=False)
    labels, clusters, nmi, kh = res[0], res[1], res[4], res[6]
    big = (np.bincount(y.astype(int))[1:] > 50).sum()
    assert big - 1 <= len(clusters) <= 7
    assert nmi[-1] > 0.9   # random 2-D components may overlap; the docs run reports 1.0 on its own dataset
    assert len(labels) == 10 ** 4 and labels.min() == 1 and labels.max() == len(clusters)
    assert kh[0] >= 1 and kh[-1] == len(clusters)


def test_module_random_mess(host):
    """test/module_tests.jl:36-47: N=10^5, D=3, 10 components, alpha=1e21 via dp_parallel; asserts K > 1."""
    x, labels, _, _ = host.generate_gaussian_data(10 ** 5, 3, 10, 100.0, seed=12345)
    hyper = host.niw_hyperparams(1.0, np.zeros(3), 5, np.eye(3))
    dp = host.dp_parallel(x, hyper, np.float32(1e21), 100, 1, None, False, False, 15, labels)
    assert dp[0].num_clusters > 1
    assert len(dp) == 5 and len(dp[1]) == 100


def test_fit_d64_recovers_components(host):
    x, y, _, _ = host.generate_gaussian_data(60000, 64, 8, 100.0, seed=3)
    res = host.fit(x, 10.0, iters=60, burnout=8, gt=y, seed=7, verbose=False)
    assert res[4][-1] > 0.98
    assert 8 <= len(res[1]) <= 10


def test_nccl_comm_path_single_rank(host):
    """The multi-GPU exchange path (device-side packed buffer -> RCCL all-reduce -> host) with a 1-rank group:
    must give exactly the LocalComm result (same seed, same data)."""
    import os
    import torch
    import torch.distributed as dist
    from dpmmsubclusters_jl_amd.host.comm import TorchDistComm
    x, y, _, _ = host.generate_gaussian_data(20000, 8, 4, 100.0, seed=9)
    ref = host.fit(x, 10.0, iters=30, burnout=5, seed=77, verbose=False)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        comm = TorchDistComm()
        assert comm.backend == "nccl"
        got = host.fit(x, 10.0, iters=30, burnout=5, seed=77, verbose=False, comm=comm)
    finally:
        dist.destroy_process_group()
    assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[7], got[7]) and ref[6] == got[6]


def test_predict_matches_reference_semantics(host):
    """test/module_tests.jl:25-28: predict(model, data)[1] == labels on the deterministic point-mass problem, and the
    posterior-predictive table equals scipy's multivariate-t closed form (priors/niw.jl:68-76)."""
    from oracle import oracle as orc
    data = np.zeros((2, 1000), np.float32)
    data[:, 0:250] = [[-1], [-1]]; data[:, 250:500] = [[-1], [1]]
    data[:, 500:750] = [[1], [-1]]; data[:, 750:1000] = [[1], [1]]
    res = host.fit(data, 100.0, iters=200, seed=123456789, burnout=15, verbose=False)
    labels, model = res[0], res[-1]
    preds, probs = host.predict(model, data)
    assert np.array_equal(preds, labels)                               # :25-28
    assert probs.shape == (1000, 4) and np.allclose(probs.sum(1), 1, atol=1e-5)
    # closed-form check on a generic problem
    x, y, _, _ = host.generate_gaussian_data(5000, 5, 3, 50.0, seed=2)
    res = host.fit(x, 10.0, iters=40, burnout=5, seed=3, verbose=False)
    s = res[-1].sampler
    preds, probs = host.predict(res[-1], x)
    w = s.points_count + s.alpha; w = w / w.sum()
    tab = np.empty((5000, s.K))
    for k in range(s.K):
        U = s.post["U"][3 * k]
        tab[:, k] = orc.niw_posterior_predictive(x.T, s.post["kappa"][3 * k], s.post["m"][3 * k], s.post["nu"][3 * k],
                                                 (U @ U.T) / s.post["nu"][3 * k]) + np.log(w[k])
    want = np.exp(tab - tab.max(1, keepdims=True)); want /= want.sum(1, keepdims=True)
    np.testing.assert_allclose(probs, want, atol=2e-4)
    assert (preds == tab.argmax(1) + 1).mean() > 0.999


def test_predict_multinomial(host):
    x, labels, _ = host.generate_mnmm_data(4000, 30, 4, 100, seed=8)
    hyper = host.multinomial_hyper(np.ones(30, np.float32))
    res = host.fit(x, hyper, 10.0, iters=40, burnout=5, seed=2, verbose=False)
    s = res[-1].sampler
    preds, probs = host.predict(res[-1], x)
    a = s.post["alpha"][[3 * k for k in range(s.K)]].astype(np.float64)
    w = s.points_count + s.alpha; w = w / w.sum()
    tab = x.T.astype(np.float64) @ np.log(a / a.sum(1, keepdims=True)).T + np.log(w)
    assert (preds == tab.argmax(1) + 1).mean() > 0.999
    assert (preds == res[0]).mean() > 0.95


def test_outlier_component_on_gpu(host):
    """fit(...; outlier_weight, outlier_params): cluster 1 is the fixed outlier component (local_clusters_actions.jl:42-61)."""
    import sys as _sys, os as _os
    _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
    from test_outlier_cpu import run_outlier_fit, check_outlier_result
    r, out = run_outlier_fit(host)
    check_outlier_result(r, out)


def test_predict_device_finish_equals_host_finish(host, pkg=None):
    """dpmm_predict_points (argmax + normalisation on the GPU) against the same steps done on the host from dpmm_predict's table."""
    from __graft_entry__ import load_package
    pkg = load_package()
    x, y, _, _ = host.generate_gaussian_data(7001, 6, 5, 60.0, seed=11)
    res = host.fit(x, 10.0, iters=30, burnout=5, seed=5, verbose=False)
    s = res[-1].sampler
    w = s.points_count.astype(np.float64) + s.alpha
    w = (w / w.sum()).astype(np.float32)
    X = np.ascontiguousarray(x.T.astype(np.float32))
    wk = pkg.Worker(s.prior.kind, X.shape[1], X.shape[0], device=0, seed=0)
    wk.upload_points(X)
    rows = [3 * k for k in range(s.K)]
    tab = s.prior.predictive_table(wk, s.post, rows, w).T.astype(np.float32)           # (n, K), host finish below
    lab_d, probs_d = s.prior.predictive_table(wk, s.post, rows, w, points=True)
    wk.close()
    lab_h = tab.argmax(1) + 1
    p = np.exp(tab - tab.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    assert np.array_equal(lab_d, lab_h)
    np.testing.assert_allclose(probs_d, p, rtol=2e-6, atol=1e-7)
    assert probs_d.shape == (X.shape[0], s.K) and lab_d.dtype == np.int64
