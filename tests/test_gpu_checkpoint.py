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


def test_resume_right_after_an_accepted_merge_with_the_device_master(pkg, host, tmp_path):
    """An accepted merge rebuilds the merged slots on the HOST, so the running chain makes its next parameter draws there (host
    streams, not the device's): a checkpoint written at the end of that very step must resume onto the same draws (ADVICE r2;
    counters[6] carries the flag).  The run starts from 12 initial clusters on 3 components so that merges are accepted; the
    checkpoint is taken at the first iteration whose cluster count dropped."""
    import importlib
    ckpt = importlib.import_module("dpmmsubclusters_jl_amd.host.checkpoint")
    D, N = 64, 6000
    X, y = host.gaussian_mixture_shard(N, D, 3, 100.0, 77, 0, N)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))

    def chain():
        wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=3)
        wk.upload_points(X)
        s = host.DPMMSampler(wk, prior, 10.0, N, 3, burnout=3)
        s.init_first_clusters(12)
        return wk, s

    wk, s = chain()
    ks, merge_at, saved = [s.K], None, None
    for it in range(1, 41):
        s.group_step(False, False)
        ks.append(s.K)
        if merge_at is None and ks[-1] < ks[-2]:
            merge_at = it
            flag = int(s.model.get("counters")[6])
            saved = ckpt.sampler_state(s) if hasattr(ckpt, "sampler_state") else {f: s.model.get(f) for f in ckpt._STATE_FIELDS}
            saved["K"] = s.K
            saved["labels"], saved["sub"] = wk.get_labels()
        if merge_at is not None and it == merge_at + 3:
            break
    assert merge_at is not None, ks
    assert flag & 1, "the step that accepted a merge leaves the next draws to the host"
    want = wk.get_labels()
    wk.close()
    # resume from the state saved right after the merge step
    wk2, s2 = chain()
    wk2.set_labels(saved["labels"], saved["sub"])
    s2.model.set("K", saved["K"])
    wk2.set_num_clusters(saved["K"])
    for f in ckpt._STATE_FIELDS:
        s2.model.set(f, saved[f])
    for _ in range(3):
        s2.group_step(False, False)
    got = wk2.get_labels()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    wk2.close()
