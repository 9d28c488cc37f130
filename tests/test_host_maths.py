"""Value-level pins of the PRODUCT's host maths (libdpmmhost.so + host/priors.py) against the reference's
golden vectors (tests/golden/*.npz, cut from the reference's own checkpoints) and against the CPU oracle's
restatement of the same formulas.  Runs everywhere (no GPU).

Reference functions under test (paths relative to the reference checkout):
    calc_posterior            src/priors/niw.jl:20-31, src/priors/multinomial_prior.jl:16-21
    log_marginal_likelihood   src/priors/niw.jl:53-62, src/priors/multinomial_prior.jl:34-39
    log_multivariate_gamma    src/utils.jl:66-72 (Float32 accumulator quirk, opt-in here)
    sample_distribution       src/priors/niw.jl:34-40, src/priors/multinomial_prior.jl:23-25
    should_merge! pooled stats src/shared_actions.jl:21-27
"""
import importlib

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import oracle as orc


@pytest.fixture(scope="module")
def host():
    load_package()
    return importlib.import_module("dpmmsubclusters_jl_amd.host")


def _niw_golden(golden_dir):
    g = np.load(f"{golden_dir}/niw_golden.npz")
    return g, (float(g["prior_kappa"]), g["prior_m"], float(g["prior_nu"]), g["prior_psi"])


def test_niw_posterior_matches_reference_checkpoint(host, golden_dir):
    """niw_hyperparams.posterior (native dpmmh_niw_posterior) on the reference's Float64 statistics reproduces the kappa, nu,
    m, psi the reference stored in examples/save_load_model/checkpoint__50.jld2."""
    g, (k0, m0, v0, p0) = _niw_golden(golden_dir)
    prior = host.niw_hyperparams(k0, m0, v0, p0)
    post = prior.posterior(g["counts"].astype(np.float64), g["points_sum"], g["S"])
    assert np.array_equal(post["kappa"], g["kappa"]) and np.array_equal(post["nu"], g["nu"])      # exact
    np.testing.assert_allclose(post["m"], g["m"], rtol=1e-9, atol=1e-12)
    psi = np.einsum("kab,kcb->kac", post["U"], post["U"]) / post["nu"][:, None, None]              # nu psi = U U'
    np.testing.assert_allclose(psi, g["psi"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(post["logdet_psi"], np.linalg.slogdet(g["psi"])[1], rtol=0, atol=1e-6)
    # U is upper triangular (reverse Cholesky): the draw code relies on it
    assert np.all(np.tril(post["U"], -1) == 0)


def _random_stats(rng, n, D, scale=3.0):
    N = rng.integers(0, 400, n).astype(np.float64)
    N[0] = 0                                   # empty statistic set -> prior (niw.jl:21-23)
    sums = np.zeros((n, D)); S = np.zeros((n, D, D))
    for i in range(n):
        if N[i] == 0:
            continue
        x = rng.normal(size=(int(N[i]), D)) * scale + rng.normal(size=D) * 5
        sums[i] = x.sum(0); S[i] = x.T @ x
    return N, sums, S


@pytest.mark.parametrize("D", [2, 7, 64])
def test_niw_posterior_and_log_marginal_vs_oracle(host, D):
    rng = np.random.default_rng(100 + D)
    k0, v0 = 1.5, D + 3.0
    m0 = rng.normal(size=D); A = rng.normal(size=(D, D)); p0 = A @ A.T / D + np.eye(D)
    prior = host.niw_hyperparams(k0, m0, v0, p0)
    N, sums, S = _random_stats(rng, 9, D)
    post = prior.posterior(N, sums, S)
    L = prior.log_marginal(post, N)
    Lq = prior.log_marginal(post, N, f32_quirk=True)
    for i in range(len(N)):
        kp, mp, vp, pp = orc.niw_calc_posterior(k0, m0, v0, p0, N[i], sums[i], S[i])
        assert post["kappa"][i] == kp and post["nu"][i] == vp
        np.testing.assert_allclose(post["m"][i], mp, rtol=1e-12, atol=1e-12)
        psi = post["U"][i] @ post["U"][i].T / post["nu"][i]
        np.testing.assert_allclose(psi, pp, rtol=1e-10, atol=1e-10)
        want = orc.niw_log_marginal((k0, m0, v0, p0), (kp, mp, vp, pp), N[i], D, f32_quirk=False)
        assert abs(L[i] - want) <= 1e-9 * max(1.0, abs(want)), (i, L[i], want)
        # the reference-compatible switch: lnGamma_D accumulated in a Float32 local (utils.jl:66-72)
        want_q = orc.niw_log_marginal((k0, m0, v0, p0), (kp, mp, vp, pp), N[i], D, f32_quirk=True)
        assert abs(Lq[i] - want_q) <= 1e-9 * max(1.0, abs(want_q)) + 1e-6 * abs(want_q - want), (i, Lq[i], want_q)
    assert L[0] == pytest.approx(0.0, abs=1e-9)      # no data: marginal likelihood 1


def test_niw_log_marginal_pairs_vs_oracle(host):
    """Pooled statistics of cluster pairs (shared_actions.jl:22-27): log_marginal_pairs == oracle on the summed stats."""
    rng = np.random.default_rng(7)
    D = 5
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3.0, np.eye(D))
    N, sums, S = _random_stats(rng, 8, D)
    pairs = np.array([(a, b) for a in range(8) for b in range(a + 1, 8)])
    got = prior.log_marginal_pairs(pairs, dict(N=N, sums=sums, S=S))
    for p, (a, b) in enumerate(pairs):
        Np, sp, Sp = N[a] + N[b], sums[a] + sums[b], S[a] + S[b]
        post = orc.niw_calc_posterior(1.0, np.zeros(D), D + 3.0, np.eye(D), Np, sp, Sp)
        want = orc.niw_log_marginal((1.0, np.zeros(D), D + 3.0, np.eye(D)), post, Np, D, f32_quirk=False)
        assert abs(got[p] - want) <= 1e-9 * max(1.0, abs(want)), (a, b)


def test_multinomial_posterior_bit_exact_and_log_marginal(host, golden_dir):
    g = np.load(f"{golden_dir}/mnm_golden.npz")
    prior = host.multinomial_hyper(g["prior_alpha"])
    sums = g["points_sum"].astype(np.float64)
    N = sums.sum(1) / 100.0
    post = prior.posterior(N, sums)
    assert post["alpha"].dtype == np.float32 and np.array_equal(post["alpha"], g["post_alpha"])     # multinomial_prior.jl:16-21
    L = prior.log_marginal(post, N)
    for i in range(len(N)):
        assert abs(L[i] - orc.mult_log_marginal(g["prior_alpha"], g["post_alpha"][i])) < 1e-8
    # pooled pairs
    pairs = np.array([[1, 2], [4, 5]])
    Lp = prior.log_marginal_pairs(pairs, dict(N=N, sums=sums))
    for p, (a, b) in enumerate(pairs):
        pa = orc.mult_calc_posterior(g["prior_alpha"], N[a] + N[b], (sums[a] + sums[b]).astype(np.float32))
        assert abs(Lp[p] - orc.mult_log_marginal(g["prior_alpha"], pa)) < 1e-8
    # empty statistic set -> prior
    assert np.array_equal(prior.posterior(np.zeros(1), np.zeros((1, 100)))["alpha"][0], g["prior_alpha"])


def test_niw_sample_moments(host):
    """sample_distribution (niw.jl:34-40): Sigma ~ InvWishart(nu, nu psi), mu ~ N(m, Sigma / kappa).
    E[Sigma] = nu psi / (nu - D - 1); cov(mu) = E[Sigma] / kappa; logdet and the factor are consistent."""
    D, n = 4, 6000
    rng = np.random.default_rng(3)
    A = rng.normal(size=(D, D)); psi = A @ A.T / D + np.eye(D)
    m = rng.normal(size=D); kappa, nu = 7.0, 30.0
    prior = host.niw_hyperparams(kappa, m, nu, psi)
    post = prior.posterior(np.zeros(1), np.zeros((1, D)), np.zeros((1, D, D)))     # N = 0 -> the prior itself
    rep = {k: np.repeat(v, n, axis=0) for k, v in post.items()}
    par = prior.sample(rep, seed=99, epoch=5, ids=np.arange(n))
    R = par["R"].astype(np.float64)
    assert np.all(np.tril(R, -1) == 0)
    W = np.einsum("kji,kjl->kil", R, R)                          # Sigma^-1 = R'R
    Sig = np.linalg.inv(W)
    np.testing.assert_allclose(par["logdet"], np.linalg.slogdet(Sig)[1], rtol=0, atol=2e-4)
    ESig = nu * psi / (nu - D - 1)
    np.testing.assert_allclose(Sig.mean(0), ESig, rtol=0.06, atol=0.03)
    np.testing.assert_allclose(W.mean(0), np.linalg.inv(psi), rtol=0.06, atol=0.03)     # E[Wishart(nu, (nu psi)^-1)] = psi^-1
    mu = par["mu"].astype(np.float64)
    np.testing.assert_allclose(mu.mean(0), m, atol=4 * np.sqrt(np.diag(ESig).max() / kappa / n))
    np.testing.assert_allclose(np.cov(mu.T), ESig / kappa, rtol=0.1, atol=0.02)
    # counter-based: the same (seed, epoch, id) gives the same draw whatever the batch / thread count
    par2 = prior.sample({k: v[10:20] for k, v in rep.items()}, seed=99, epoch=5, ids=np.arange(10, 20), nthreads=1)
    assert np.array_equal(par2["R"], par["R"][10:20]) and np.array_equal(par2["mu"], par["mu"][10:20])
    # pre-generated noise path == direct path
    noise = prior.draw_noise(n, 99, 5)
    par3 = prior.sample(rep, seed=99, epoch=5, ids=np.arange(n), noise=noise)
    assert np.array_equal(par3["R"], par["R"]) and np.array_equal(par3["mu"], par["mu"])


def test_dirichlet_log_moments(host):
    """log.(rand(Dirichlet(alpha'))) (multinomial_prior.jl:23-25): exp sums to one, means alpha / sum(alpha)."""
    D, n = 6, 20000
    alpha = np.array([0.3, 1.0, 2.5, 7.0, 0.05, 12.0], np.float32)
    prior = host.multinomial_hyper(alpha)
    post = dict(alpha=np.repeat(alpha[None], n, axis=0))
    lp = prior.sample(post, seed=5, epoch=1, ids=np.arange(n))["logp"].astype(np.float64)
    p = np.exp(lp)
    np.testing.assert_allclose(p.sum(1), 1.0, atol=1e-5)
    a0 = float(alpha.sum())
    np.testing.assert_allclose(p.mean(0), alpha / a0, atol=4e-3)
    var = alpha / a0 * (1 - alpha / a0) / (a0 + 1)
    np.testing.assert_allclose(p.var(0), var, rtol=0.1, atol=2e-4)
