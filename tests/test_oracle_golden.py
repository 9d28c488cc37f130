"""Pins the CPU oracle against the reference's own artefacts (tests/golden/*.npz, extracted by
tests/golden/make_golden.py from the reference's committed checkpoints) and against closed forms
that are independent of the reference (scipy)."""
import numpy as np
import pytest
from scipy import stats
from scipy.special import gammaln, multigammaln

from oracle import oracle as orc


def _bins(K):
    return [(k, w) for k in range(K) for w in range(3)]


def test_mult_suffstats_and_posterior_bit_exact(golden_dir):
    g = np.load(f"{golden_dir}/mnm_golden.npz")
    X = np.ascontiguousarray(g["X"], np.float32)  # (n, D): row = point
    N, s = orc.suffstats_mult(X, 100, g["labels"], g["sub"], 2)
    for i, (k, w) in enumerate(_bins(2)):
        assert np.array_equal(s[k, w], g["points_sum"][i]), (k, w)  # priors/multinomial_prior.jl:27-32
        post = orc.mult_calc_posterior(g["prior_alpha"], N[k, w], s[k, w])
        assert np.array_equal(post, g["post_alpha"][i])  # priors/multinomial_prior.jl:16-21
    assert N[:, 0].tolist() == [463, 537] and N[:, 1].tolist() == [251, 221] and N[:, 2].tolist() == [212, 316]


def test_niw_suffstats_and_posterior_golden(golden_dir):
    g = np.load(f"{golden_dir}/niw_golden.npz")
    X64 = g["X"]
    # the golden run used Float64 data; the oracle consumes Float32 points (ds.jl:53), so feed the
    # f32-rounded data and compare with golden values recomputed tolerance: rounding of inputs (6e-8 rel).
    X = np.ascontiguousarray(X64, np.float32)
    N, s, S = orc.suffstats_niw(X, 2, g["labels"], g["sub"], 5)
    prior = (float(g["prior_kappa"]), g["prior_m"], float(g["prior_nu"]), g["prior_psi"])
    for i, (k, w) in enumerate(_bins(5)):
        assert N[k, w] == g["counts"][i]
        np.testing.assert_allclose(s[k, w], g["points_sum"][i], rtol=0, atol=5e-7 * np.abs(X64).max() * max(N[k, w], 1))
        np.testing.assert_allclose(S[k, w], g["S"][i], rtol=2e-6, atol=1e-4)
        # posterior from the GOLDEN Float64 statistics must reproduce the golden posterior (priors/niw.jl:20-31)
        kp, mp, vp, pp = orc.niw_calc_posterior(*prior, g["counts"][i], g["points_sum"][i], g["S"][i])
        assert kp == g["kappa"][i] and vp == g["nu"][i]
        np.testing.assert_allclose(mp, g["m"][i], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(pp, g["psi"][i], rtol=0, atol=3e-7)


def test_niw_suffstats_f64_exact_on_f32_inputs():
    rng = np.random.default_rng(0)
    n, D, K = 500, 5, 3
    X = rng.normal(size=(n, D)).astype(np.float32)
    lab = rng.integers(1, K + 1, n); sub = rng.integers(1, 3, n)
    N, s, S = orc.suffstats_niw(X, D, lab, sub, K)
    Xd = X.astype(np.float64)
    for k in range(K):
        for w, m in enumerate((lab == k + 1, (lab == k + 1) & (sub == 1), (lab == k + 1) & (sub == 2))):
            assert N[k, w] == m.sum()
            np.testing.assert_allclose(s[k, w], Xd[m].sum(0), rtol=1e-13, atol=1e-12)
            np.testing.assert_allclose(S[k, w], Xd[m].T @ Xd[m], rtol=1e-13, atol=1e-12)
            assert np.array_equal(S[k, w], S[k, w].T)


def test_niw_loglik_matches_scipy_up_to_the_normaliser_quirk():
    # mv_gaussian.jl:24 uses length(Sigma) = D^2 in the normaliser: loglik_ref = logpdf - (D^2 - D)/2 log(2 pi)
    rng = np.random.default_rng(1)
    D, n = 6, 200
    A = rng.normal(size=(D, D)); Sigma = A @ A.T + D * np.eye(D)
    mu = rng.normal(size=D) * 3
    X = (rng.multivariate_normal(mu, Sigma, size=n)).astype(np.float32)
    invS = np.linalg.inv(Sigma)
    logdet = np.linalg.slogdet(Sigma)[1]
    got = orc.niw_loglik_ref(X, D, mu, invS.T.ravel(), logdet)
    want = stats.multivariate_normal(mu.astype(np.float32), Sigma).logpdf(X.astype(np.float64)) - 0.5 * (D * D - D) * np.log(2 * np.pi)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-4)
    got64 = orc.niw_loglik_f64(X, D, mu, invS.T.ravel(), logdet)
    np.testing.assert_allclose(got64, want, rtol=1e-6, atol=1e-5)


def test_mult_loglik():
    rng = np.random.default_rng(2)
    D, n = 50, 64
    X = rng.multinomial(30, np.ones(D) / D, size=n).astype(np.float32)
    logp = np.log(rng.dirichlet(np.ones(D))).astype(np.float32)
    np.testing.assert_allclose(orc.mult_loglik_ref(X, D, logp), X.astype(np.float64) @ logp.astype(np.float64), rtol=1e-5)


def test_exp_det_accuracy():
    x = -np.abs(np.random.default_rng(3).normal(size=2000) * 20).astype(np.float32)
    got = orc.exp_det(x)
    want = np.exp(x.astype(np.float64))
    ok = x >= -86
    np.testing.assert_allclose(got[ok], want[ok], rtol=4e-7)
    assert np.all(got[~ok] == 0)
    assert orc.exp_det(np.float32(0))[0] == 1.0
    assert orc.exp_det(np.float32(-np.inf))[0] == 0.0


def test_sample_log_cat_semantics():
    # utils.jl:19-31 + StatsBase inverse-CDF scan
    parr = np.log(np.array([[0.2], [0.5], [0.3]], np.float32))
    for u, want in ((0.0, 1), (0.19, 1), (0.21, 2), (0.69, 2), (0.71, 3), (0.999, 3)):
        assert orc.sample_log_cat(parr, np.array([u], np.float32))[0] == want
    # NaN -> -Inf ; all -Inf row -> 1 ; shift invariance
    p = np.array([[np.nan, -np.inf, 5.0], [0.0, -np.inf, 5.0], [np.nan, -np.inf, -1e30]], np.float32)
    lab = orc.sample_log_cat(p, np.array([0.5, 0.9, 0.3], np.float32))
    assert lab.tolist() == [2, 1, 1] or lab.tolist() == [2, 1, 2]
    assert lab[1] == 1
    # frequencies
    n = 200000
    w = np.array([0.1, 0.2, 0.3, 0.4])
    parr = np.repeat(np.log(w).astype(np.float32)[:, None] + 123.0, n, axis=1).astype(np.float32)
    u, _ = orc.uniforms(7, 0, 0, 0, n)
    lab = orc.sample_log_cat(parr, u)
    freq = np.bincount(lab, minlength=5)[1:] / n
    assert np.abs(freq - w).max() < 5e-3


def test_argmax_first_max_wins():
    p = np.array([[1.0, 3.0], [2.0, 3.0], [2.0, 1.0]], np.float32)
    assert orc.argmax_rows(p).tolist() == [2, 1]


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    assert orc.philox(0, 0, 0, 0) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    lo = 0xFFFFFFFFFFFFFFFF
    out = (np.zeros(4, np.uint32))
    import ctypes
    o = (ctypes.c_uint32 * 4)()
    orc.lib().orc_philox(ctypes.c_uint64(lo), ctypes.c_uint64(lo), ctypes.c_uint32(0xFFFFFFFF), ctypes.c_uint32(0xFFFFFFFF), o)
    assert [int(v) for v in o] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]


def test_uniform_stream_is_uniform_and_index_keyed():
    u0, u1 = orc.uniforms(123456789, 3, 0, 1000, 100000)
    assert 0 <= u0.min() and u0.max() < 1 and abs(u0.mean() - 0.5) < 5e-3 and abs(np.corrcoef(u0, u1)[0, 1]) < 0.02
    a, _ = orc.uniforms(123456789, 3, 0, 1500, 10)
    assert np.array_equal(a, u0[500:510])  # counter = global index -> shard independent


def test_relabel_ops():
    rng = np.random.default_rng(5)
    n = 5000
    lab = rng.integers(1, 6, n).astype(np.int64); sub = rng.integers(1, 3, n).astype(np.int64)
    lab0, sub0 = lab.copy(), sub.copy()
    orc.split_relabel(lab, sub, [2, 4], [6, 7], seed=9, epoch=1)
    assert np.all(lab[(lab0 == 2) & (sub0 == 2)] == 6) and np.all(lab[(lab0 == 2) & (sub0 == 1)] == 2)
    assert np.all(lab[(lab0 == 4) & (sub0 == 2)] == 7)
    untouched = ~np.isin(lab0, [2, 4])
    assert np.array_equal(lab[untouched], lab0[untouched]) and np.array_equal(sub[untouched], sub0[untouched])
    assert set(np.unique(sub[~untouched])) == {1, 2} and abs((sub[~untouched] == 1).mean() - 0.5) < 0.05
    # merge: (1 <- 3) then (1 <- 5) in order (local_clusters_actions.jl:293-304)
    l1, s1 = lab.copy(), sub.copy()
    orc.merge_relabel(l1, s1, [1, 1], [3, 5])
    assert not np.any(np.isin(l1, [3, 5]))
    assert np.all(s1[lab == 5] == 2) and np.all(s1[lab == 3] == 1) and np.all(s1[lab == 1] == 1)
    # remove empty: counts with zeros at (1-based) 3 and 5 -> labels above shift down
    l2 = l1.copy()
    cnt = np.bincount(l1, minlength=8)[1:8]
    orc.remove_empty(l2, cnt)
    expect = {1: 1, 2: 2, 4: 3, 6: 4, 7: 5}
    assert all(np.all(l2[l1 == a] == b) for a, b in expect.items())
    # reset
    s3 = s1.copy()
    orc.reset_sub(l1, s3, [2], seed=9, epoch=2)
    assert np.array_equal(s3[l1 != 2], s1[l1 != 2]) and abs((s3[l1 == 2] == 1).mean() - 0.5) < 0.06
    lab_i, sub_i = orc.init_labels(20000, 7, seed=1, epoch=0)
    assert lab_i.min() == 1 and lab_i.max() == 7 and abs((sub_i == 1).mean() - 0.5) < 0.02


def test_niw_log_marginal_matches_textbook():
    # priors/niw.jl:53-62 == standard NIW evidence with IW scale nu*psi
    rng = np.random.default_rng(6)
    D, n = 3, 40
    X = rng.normal(size=(n, D)) * 2 + 1
    k0, m0, v0, p0 = 1.0, np.zeros(D), 5.0, np.eye(D)
    post = orc.niw_calc_posterior(k0, m0, v0, p0, n, X.sum(0), X.T @ X)
    got = orc.niw_log_marginal((k0, m0, v0, p0), post, n, D, f32_quirk=False)
    k1, m1, v1, p1 = post
    L0, L1 = v0 * p0, v1 * p1
    want = (-n * D / 2 * np.log(np.pi) + multigammaln(v1 / 2, D) - multigammaln(v0 / 2, D)
            + v0 / 2 * np.linalg.slogdet(L0)[1] - v1 / 2 * np.linalg.slogdet(L1)[1] + D / 2 * np.log(k0 / k1))
    assert abs(got - want) < 1e-8
    # posterior scale identity: L1 = L0 + sum (x-xbar)(x-xbar)' + k0 n/(k0+n) (xbar-m0)(xbar-m0)'
    xb = X.mean(0)
    Lw = L0 + (X - xb).T @ (X - xb) + k0 * n / (k0 + n) * np.outer(xb - m0, xb - m0)
    np.testing.assert_allclose(L1, Lw, rtol=1e-10, atol=1e-10)
    quirk = orc.niw_log_marginal((k0, m0, v0, p0), post, n, D, f32_quirk=True)
    assert abs(quirk - got) < 1e-2


def test_mult_log_marginal_is_dirichlet_multinomial_evidence():
    a = np.ones(5, np.float32); cnt = np.array([3, 0, 2, 5, 1], np.float32)
    got = orc.mult_log_marginal(a, orc.mult_calc_posterior(a, cnt.sum(), cnt))
    want = gammaln(5) - gammaln(5 + 11) + np.sum(gammaln(1 + cnt.astype(np.float64)))
    assert abs(got - want) < 1e-10


def test_sweep_c_vs_numpy_restatement():
    rng = np.random.default_rng(8)
    D, n, K = 4, 3000, 3
    mus = rng.normal(size=(3 * K, D)) * 4
    for k in range(K):
        mus[3 * k + 1] = mus[3 * k] + 0.5; mus[3 * k + 2] = mus[3 * k] - 0.5
    Sig = np.stack([np.eye(D) * (0.5 + rng.random()) for _ in range(3 * K)])
    invS = np.linalg.inv(Sig); logdet = np.linalg.slogdet(Sig)[1]
    z = rng.integers(0, K, n)
    X = (mus[3 * z] + rng.normal(size=(n, D))).astype(np.float32)
    logw = np.log(np.ones(K) / K); loglr = np.log(np.full((K, 2), 0.5))
    lab, sub, parr = orc.sweep_niw(X, D, mus, invS.reshape(3 * K, -1), logdet, logw, loglr, seed=11, epoch=4, want_parr=True)
    u0, u1 = orc.uniforms(11, 4, 0, 0, n)
    lab2, sub2, (N2, s2, S2) = orc.sweep_numpy_niw(X, D, mus.astype(np.float32), invS.reshape(3 * K, -1).astype(np.float32),
                                                     logdet.astype(np.float32), logw.astype(np.float32), loglr.astype(np.float32), u0, u1)
    assert (lab != lab2).mean() < 2e-3 and (sub[lab == lab2] != sub2[lab == lab2]).mean() < 5e-3
    N, s, S = orc.suffstats_niw(X, D, lab2, sub2, K)
    np.testing.assert_allclose(N, N2); np.testing.assert_allclose(s, s2, rtol=1e-12, atol=1e-9); np.testing.assert_allclose(S, S2, rtol=1e-12, atol=1e-9)
    assert (lab == z + 1).mean() > 0.9
