"""Size-independent properties at BASELINE.json's full sizes (C3 per-node size N = 10^7, D = 64 NIW; C4 N = 10^6, D = 1000
Multinomial): the oracle cannot run these sizes in seconds, so the checks are invariants of the path itself --
conservation of counts and moments, shard invariance, bitwise reproducibility, idempotence of the relabel operations."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="module")
def host(pkg):
    import importlib
    return importlib.import_module("dpmmsubclusters_jl_amd.host")


def _niw_params(host, X, y, K, D, seed):
    """One posterior draw per true component (cluster and both sub-clusters), as a sweep would see them."""
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    rng = np.random.default_rng(seed)
    idx = rng.choice(len(X), 200000, replace=False)
    N = np.zeros(3 * K); sums = np.zeros((3 * K, D)); S = np.zeros((3 * K, D, D))
    for k in range(K):
        pts = X[idx][y[idx] == k + 1].astype(np.float64)
        for w in range(3):
            sel = pts if w == 0 else pts[w - 1::2]
            N[3 * k + w] = len(sel); sums[3 * k + w] = sel.sum(0); S[3 * k + w] = sel.T @ sel
    post = prior.posterior(N, sums, S, nthreads=8)
    par = prior.sample(post, 7, 1, np.arange(3 * K), nthreads=8)
    return prior, par


def _windows(y, n, width):
    """Three windows of `width` points: the start, one straddling a component boundary, the end."""
    b = int(np.flatnonzero(np.diff(y) != 0)[len(np.flatnonzero(np.diff(y) != 0)) // 2]) + 1 if np.any(np.diff(y) != 0) else n // 2
    mid = max(0, min(n - width, b - width // 2))
    return [(0, width), (mid, mid + width), (n - width, n)]


def _check_niw_windows(host, X, y, par, lr, w, lab, sub, seed, epoch, width, first=0):
    """The oracle restatement of sample_labels_worker! / create_subclusters_labels! on windows of the full-size run: draws are
    independent per point given the parameters (index-keyed uniforms), so any window can be checked exactly as the small
    problems are -- labels equal up to counted boundary flips (<= 1e-5 of the window, at least 1), sub-labels up to 1e-4 (at least 2); counts printed."""
    K = len(w); D = X.shape[1]
    inv, _ = host.native.niw_expand(par["R"], want_sigma=False)
    invS = inv.reshape(3 * K, -1).astype(np.float32)
    for lo, hi in _windows(y, len(X), width):
        olab, osub = orc.sweep_niw(np.ascontiguousarray(X[lo:hi]), D, par["mu"], invS, par["logdet"], np.log(w), np.log(lr), seed=seed,
                                   epoch=epoch, first_idx=first + lo)
        flips = int((lab[lo:hi] != olab).sum())
        assert flips <= max(1, int(1e-5 * (hi - lo))), (lo, hi, flips)
        same = lab[lo:hi] == olab
        sflips = int((sub[lo:hi][same] != osub[same]).sum())
        print(f"window [{lo}, {hi}): label flips vs oracle {flips}, sub-label flips {sflips}")
        assert sflips <= max(2, int(1e-4 * (hi - lo))), (lo, hi, sflips)


def test_c3_full_size_niw_properties(pkg, host):
    N, D, K = 10 ** 7, 64, 32
    X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
    prior, par = _niw_params(host, X, y, K, D, 3)
    lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
    sub0 = 1 + (np.arange(N) & 1)

    def run(lo, hi, seed=99):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, hi - lo, first_index=lo, device=0, seed=seed)
        wk.upload_points(X[lo:hi])
        wk.set_labels(y[lo:hi], sub0[lo:hi])
        prior.upload(wk, par, lr, w)
        wk.sweep(5)
        lab, sub = wk.get_labels()
        packed = wk.suffstats_packed(None)
        return wk, lab, sub, packed

    wk, lab, sub, packed = run(0, N)
    assert lab.min() >= 1 and lab.max() <= K and set(np.unique(sub)) <= {1, 2}
    # the oracle on three 20 000-point windows of the SECOND sweep: previous labels, the bin-sorted processing order and the
    # cluster screening (which removes 31 of 32 clusters on this data) are all active then
    prior.upload(wk, par, lr, w)
    wk.sweep(6)
    lab6, sub6 = wk.get_labels()
    _check_niw_windows(host, X, y, par, lr, w, lab6, sub6, seed=99, epoch=6, width=20000)
    wk.set_labels(lab, sub)
    assert (lab == y).mean() > 0.999                              # well separated data: the sweep keeps the components
    Nk, sums, S = wk.unpack(packed, K)
    # conservation: counts exactly, first and second moments against an independent Float64 pass over X
    assert Nk[:, 0].sum() == N and np.array_equal(Nk[:, 0], Nk[:, 1] + Nk[:, 2])
    assert np.array_equal(Nk[:, 0], np.bincount(lab, minlength=K + 1)[1:])
    col = np.zeros(D); sq = 0.0
    for a in range(0, N, 10 ** 6):
        blk = X[a:a + 10 ** 6].astype(np.float64)
        col += blk.sum(0); sq += float((blk * blk).sum())
    np.testing.assert_allclose(sums[:, 0].sum(0), col, rtol=1e-10, atol=1e-6)
    np.testing.assert_allclose(np.trace(S[:, 0].sum(0)), sq, rtol=1e-10)
    assert np.allclose(S[:, 0], np.transpose(S[:, 0], (0, 2, 1)))
    # bitwise reproducibility of labels and statistics (no atomics, fixed reduction order)
    wk.set_labels(y, sub0); wk.sweep(5)
    lab2, sub2 = wk.get_labels()
    assert np.array_equal(lab, lab2) and np.array_equal(sub, sub2)
    assert np.array_equal(wk.suffstats_packed(None), packed)
    # idempotence of the relabel operations
    counts = Nk[:, 0].astype(np.int64)
    wk.remove_empty(counts)                                       # nothing is empty: labels must not move
    assert np.array_equal(wk.get_labels()[0], lab)
    wk.merge(np.array([1]), np.array([2])); wk.set_num_clusters(K)
    lm, sm = wk.get_labels()
    assert (lm == 2).sum() == 0 and (lm == 1).sum() == counts[0] + counts[1]
    assert np.array_equal(sm[lab == 1], np.ones((lab == 1).sum(), np.int64)) and np.array_equal(sm[lab == 2], np.full((lab == 2).sum(), 2))
    wk.close()
    # shard invariance at full size: two shards (as two ranks would hold them) give the same labels and the same total statistics
    h = N // 2 + 12345
    wa, la, sa, pa = run(0, h)
    wb, lb, sb, pb = run(h, N)
    assert np.array_equal(np.concatenate([la, lb]), lab) and np.array_equal(np.concatenate([sa, sb]), sub)
    Na, suma, Sa = wa.unpack(pa, K); Nb, sumb, Sb = wb.unpack(pb, K)
    assert np.array_equal(Na + Nb, Nk)
    np.testing.assert_allclose(suma + sumb, sums, rtol=1e-12, atol=1e-7)
    np.testing.assert_allclose(Sa + Sb, S, rtol=1e-12, atol=1e-6)
    wa.close(); wb.close()


def test_c4_full_size_multinomial_properties(pkg, host):
    """BASELINE config 4 at full size: Multinomial D = 1000, N = 1e6, K = 32, data from the reference generator's recipe (generate_mnmm_data,
    data_generators.jl:59-72: per component Dirichlet(a), a_d ~ U{1..20} with one entry ~ U{30..100}; 100 trials per document) -- the data
    the bench leg uses.  Conservation (exact integers), and the oracle on three 20 000-point windows with the budgets DESIGN section 5 states
    for every sweep test: labels max(1, 1e-5 w), sub-labels max(2, 1e-4 w); the measured counts are printed."""
    N, D, K = 10 ** 6, 1000, 32
    x, labels, clusters = host.generate_mnmm_data(N, D, K, 100, seed=4)
    X = np.ascontiguousarray(x.T); del x
    z = labels - 1
    P = clusters.T
    logp = np.log(np.maximum(np.repeat(P, 3, axis=0), 1e-30)).astype(np.float32)
    wk = pkg.Worker(pkg.PRIOR_MULT, D, N, device=0, seed=5)
    wk.upload_points(X)
    wk.set_params_mult(logp, np.full((K, 2), 0.5, np.float32), np.full(K, 1.0 / K, np.float32))
    wk.init_labels(K, 1)
    wk.sweep(2)
    lab, sub = wk.get_labels()
    assert (lab == z + 1).mean() > 0.999 and set(np.unique(sub)) <= {1, 2}
    Nk, sums = wk.unpack(wk.suffstats_packed(None), K)[:2]
    assert Nk[:, 0].sum() == N and np.array_equal(Nk[:, 0], np.bincount(lab, minlength=K + 1)[1:])
    assert np.array_equal(sums[:, 0].sum(1), 100.0 * Nk[:, 0])              # every document has exactly 100 words: exact in Float64
    assert np.array_equal(sums[:, 0].sum(0), X.astype(np.float64).sum(0))   # counts are integers: exact
    # the oracle on three 20 000-point windows (draws are independent per point given the parameters)
    logw = np.log(np.full(K, 1.0 / K, np.float32)); loglr = np.log(np.full((K, 2), 0.5, np.float32))
    for lo, hi in ((0, 20000), (N // 2 - 10000, N // 2 + 10000), (N - 20000, N)):
        w = hi - lo
        olab, osub = orc.sweep_mult(np.ascontiguousarray(X[lo:hi]), D, logp, logw, loglr, 5, 2, lo)
        flips = int((lab[lo:hi] != olab).sum())
        same = lab[lo:hi] == olab
        sflips = int((sub[lo:hi][same] != osub[same]).sum())
        print(f"C4 window [{lo}, {hi}): label flips vs oracle {flips} (budget {max(1, int(1e-5 * w))}), sub-label flips {sflips} (budget {max(2, int(1e-4 * w))})")
        assert flips <= max(1, int(1e-5 * w)), (lo, flips)
        assert sflips <= max(2, int(1e-4 * w)), (lo, sflips)
    wk.sweep(2)
    lab2, _ = wk.get_labels()
    assert (lab2 == lab).mean() > 0.999                                      # same epoch, same parameters: the labels stay with their components
    wk.close()


def test_c5_shard_size_niw_d256(pkg, host):
    """BASELINE config 5 at its per-GPU size: NIW D = 256, n = 6.25e5 points (N = 5e6 over 8 GPUs), K = 32 -- the LDS-staged
    sweep kernel with workgroup-wide screening and the 16-block statistics kernel.  Conservation, reproducibility, and the
    oracle on three windows of the second sweep."""
    n, D, K = 625000, 256, 32
    X, y = host.gaussian_mixture_shard(n, D, K, 100.0, 12345, 0, n)
    prior, par = _niw_params(host, X, y, K, D, 3)
    lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
    sub0 = 1 + (np.arange(n) & 1)
    wk = pkg.Worker(pkg.PRIOR_NIW, D, n, first_index=0, device=0, seed=41)
    wk.upload_points(X)
    wk.set_labels(y, sub0)
    prior.upload(wk, par, lr, w)
    wk.sweep(5)
    lab, sub = wk.get_labels()
    packed = wk.suffstats_packed(None)
    assert (lab == y).mean() > 0.999 and set(np.unique(sub)) <= {1, 2}
    Nk, sums, S = wk.unpack(packed, K)
    assert Nk[:, 0].sum() == n and np.array_equal(Nk[:, 0], np.bincount(lab, minlength=K + 1)[1:])
    assert np.array_equal(Nk[:, 0], Nk[:, 1] + Nk[:, 2])
    Xd = X.astype(np.float64)
    np.testing.assert_allclose(sums[:, 0].sum(0), Xd.sum(0), rtol=1e-10, atol=1e-6)
    np.testing.assert_allclose(np.trace(S[:, 0].sum(0)), float((Xd * Xd).sum()), rtol=1e-10)
    k = int(np.argmax(Nk[:, 0]))                                     # one full second-moment matrix against numpy
    m = lab == k + 1
    np.testing.assert_allclose(S[k, 0], Xd[m].T @ Xd[m], rtol=1e-11, atol=1e-6)
    del Xd
    # second sweep: ordered processing + screening active; bitwise reproducible; oracle windows
    prior.upload(wk, par, lr, w)
    wk.sweep(6)
    lab6, sub6 = wk.get_labels()
    _check_niw_windows(host, X, y, par, lr, w, lab6, sub6, seed=41, epoch=6, width=1500)
    wk.set_labels(lab, sub); wk.suffstats_packed(None)
    prior.upload(wk, par, lr, w)
    wk.sweep(6)
    l2, s2 = wk.get_labels()
    assert np.array_equal(l2, lab6) and np.array_equal(s2, sub6)
    wk.close()


def test_c2_size_niw_d64(pkg, host):
    """BASELINE config 2 at its size (it ran only as a bench leg): NIW D = 64, N = 1e6, 32 true components, one GPU -- below 4e6
    points the sort works in tiles of 512 points (C3's N = 1e7 takes the 2048-point ones) and the statistics pass in 1024 workgroups.
    Conservation against an independent Float64 pass, the oracle on three windows of the SECOND sweep (bin-sorted visiting order, lean launch
    and screening active), bitwise reproducibility, two-shard invariance."""
    N, D, K = 10 ** 6, 64, 32
    X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 2024, 0, N)
    prior, par = _niw_params(host, X, y, K, D, 5)
    lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
    sub0 = 1 + (np.arange(N) & 1)

    def run(lo, hi, seed=77):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, hi - lo, first_index=lo, device=0, seed=seed)
        wk.upload_points(X[lo:hi])
        wk.set_timing(15)
        wk.set_labels(y[lo:hi], sub0[lo:hi])
        prior.upload(wk, par, lr, w)
        wk.sweep(5)
        lab, sub = wk.get_labels()
        packed = wk.suffstats_packed(None)
        prior.upload(wk, par, lr, w)
        wk.last_sweep_work()
        wk.sweep(6)
        lean_ms = wk.last_sweep_parts_ms()[0]
        work = wk.last_sweep_work()
        lab6, sub6 = wk.get_labels()
        return wk, lab, sub, packed, lab6, sub6, lean_ms, work

    wk, lab, sub, packed, lab6, sub6, lean_ms, work = run(0, N)
    assert lean_ms > 0.0 and work["b3_evals"] > 0                     # the second sweep ran as lean launch + list launch
    assert lab.min() >= 1 and lab.max() <= K and set(np.unique(sub)) <= {1, 2} and (lab == y).mean() > 0.999
    Nk, sums, S = wk.unpack(packed, K)
    assert Nk[:, 0].sum() == N and np.array_equal(Nk[:, 0], Nk[:, 1] + Nk[:, 2])
    assert np.array_equal(Nk[:, 0], np.bincount(lab, minlength=K + 1)[1:])
    Xd = X.astype(np.float64)
    np.testing.assert_allclose(sums[:, 0].sum(0), Xd.sum(0), rtol=1e-10, atol=1e-6)
    np.testing.assert_allclose(np.trace(S[:, 0].sum(0)), float((Xd * Xd).sum()), rtol=1e-10)
    for k in (int(np.argmax(Nk[:, 0])), int(np.argmin(Nk[:, 0]))):   # two full second-moment matrices (and their sub-cluster halves) against numpy
        for h in (0, 1, 2):
            m = (lab == k + 1) if h == 0 else ((lab == k + 1) & (sub == h))
            np.testing.assert_allclose(S[k, h], Xd[m].T @ Xd[m], rtol=1e-11, atol=1e-6)
            np.testing.assert_allclose(sums[k, h], Xd[m].sum(0), rtol=1e-11, atol=1e-7)
    del Xd
    _check_niw_windows(host, X, y, par, lr, w, lab6, sub6, seed=77, epoch=6, width=20000)
    # bitwise reproducibility of the two-sweep chain
    wk.set_labels(y, sub0); prior.upload(wk, par, lr, w); wk.sweep(5)
    l5, s5 = wk.get_labels()
    assert np.array_equal(l5, lab) and np.array_equal(s5, sub)
    assert np.array_equal(wk.suffstats_packed(None), packed)
    prior.upload(wk, par, lr, w); wk.sweep(6)
    l6, s6 = wk.get_labels()
    assert np.array_equal(l6, lab6) and np.array_equal(s6, sub6)
    wk.close()
    h = N // 3 + 77                                                   # two ragged shards: same labels, same total statistics
    wa, la, sa, pa, la6, sa6, _, _ = run(0, h)
    wb, lb, sb, pb, lb6, sb6, _, _ = run(h, N)
    assert np.array_equal(np.concatenate([la, lb]), lab) and np.array_equal(np.concatenate([sa, sb]), sub)
    assert np.array_equal(np.concatenate([la6, lb6]), lab6) and np.array_equal(np.concatenate([sa6, sb6]), sub6)
    Na, suma, Sa = wa.unpack(pa, K); Nb, sumb, Sb = wb.unpack(pb, K)
    assert np.array_equal(Na + Nb, Nk)
    np.testing.assert_allclose(suma + sumb, sums, rtol=1e-12, atol=1e-7)
    np.testing.assert_allclose(Sa + Sb, S, rtol=1e-12, atol=1e-6)
    wa.close(); wb.close()


def test_c5_full_n_niw_d256(pkg, host):
    """BASELINE config 5 at its FULL N on one GPU (5e6 x 256 Float32 = 5.1 GB fits one MI355X): NIW D = 256, N = 5e6, K = 32.
    The host holds the Float32 matrix only; the independent Float64 pass runs in blocks of 5e5 points.
    Conservation against that pass, the oracle on three windows of the second sweep (bracket launch, workgroup-wide
    screening, the 16-block statistics kernel with 2048-point sort tiles), bitwise reproducibility."""
    N, D, K = 5 * 10 ** 6, 256, 32
    X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 4321, 0, N)
    prior, par = _niw_params(host, X, y, K, D, 9)
    lr = np.full((K, 2), 0.5, np.float32); w = np.full(K, 1.0 / K, np.float32)
    sub0 = 1 + (np.arange(N) & 1)
    wk = pkg.Worker(pkg.PRIOR_NIW, D, N, first_index=0, device=0, seed=43)
    wk.upload_points(X)
    wk.set_labels(y, sub0)
    prior.upload(wk, par, lr, w)
    wk.sweep(5)
    lab, sub = wk.get_labels()
    packed = wk.suffstats_packed(None)
    assert (lab == y).mean() > 0.999 and set(np.unique(sub)) <= {1, 2}
    Nk, sums, S = wk.unpack(packed, K)
    assert Nk[:, 0].sum() == N and np.array_equal(Nk[:, 0], np.bincount(lab, minlength=K + 1)[1:])
    assert np.array_equal(Nk[:, 0], Nk[:, 1] + Nk[:, 2])
    col = np.zeros(D); sq = 0.0
    k = int(np.argmin(Nk[:, 0])); Sk = np.zeros((D, D))
    for a in range(0, N, 500000):
        blk = X[a:a + 500000].astype(np.float64)
        col += blk.sum(0); sq += float((blk * blk).sum())
        m = lab[a:a + 500000] == k + 1
        if m.any():
            Sk += blk[m].T @ blk[m]
    np.testing.assert_allclose(sums[:, 0].sum(0), col, rtol=1e-10, atol=1e-6)
    np.testing.assert_allclose(np.trace(S[:, 0].sum(0)), sq, rtol=1e-10)
    np.testing.assert_allclose(S[k, 0], Sk, rtol=1e-10, atol=1e-6)
    prior.upload(wk, par, lr, w)
    wk.sweep(6)
    lab6, sub6 = wk.get_labels()
    _check_niw_windows(host, X, y, par, lr, w, lab6, sub6, seed=43, epoch=6, width=1500)
    wk.set_labels(lab, sub); wk.suffstats_packed(None)
    prior.upload(wk, par, lr, w)
    wk.sweep(6)
    l2, s2 = wk.get_labels()
    assert np.array_equal(l2, lab6) and np.array_equal(s2, sub6)
    wk.close()
