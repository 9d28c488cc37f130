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


@pytest.mark.parametrize("D", [2, 5, 64])
def test_niw_sample_law_at_small_nu(host, D):
    """The host draw at nu = D + 3 (the prior of an empty sub-cluster, where a degrees-of-freedom error of one is 1 / nu of the
    mean): Sigma^-1 = R'R ~ Wishart(nu, (nu psi)^-1), E = psi^-1, every entry to 5 standard errors over 20 000 draws (0.6 % of the
    diagonal at D = 64, 2.2 % at D = 2)."""
    n, chunk = 20000, 2000
    rng = np.random.default_rng(11 + D)
    A = rng.normal(size=(D, D)); psi = A @ A.T / D + np.eye(D)
    m = rng.normal(size=D); kappa, nu = 1.5, D + 3.0
    prior = host.niw_hyperparams(kappa, m, nu, psi)
    post = prior.posterior(np.zeros(1), np.zeros((1, D)), np.zeros((1, D, D)))
    rep = {k: np.repeat(v, chunk, axis=0) for k, v in post.items()}
    W = np.zeros((D, D))
    for c in range(n // chunk):
        par = prior.sample(rep, seed=4242, epoch=3, ids=np.arange(c * chunk, (c + 1) * chunk))
        R = par["R"].astype(np.float64)
        W += np.einsum("kji,kjl->il", R, R)
    Wm = W / n
    EW = np.linalg.inv(psi)
    se = np.sqrt((EW ** 2 + np.outer(np.diag(EW), np.diag(EW))) / nu / n)
    assert np.all(np.abs(Wm - EW) < 5 * se), np.max(np.abs(Wm - EW) / se)
    assert np.max(np.abs(np.diag(Wm) / np.diag(EW) - 1)) < 0.5 / nu


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


# --------------------------------------------------------------------------------------------- the native engine's decisions
def _engine_with_stats(host, D, K, rng, burnout=5, alpha=10.0, f32_quirk=False):
    """A dpmmh_model (host/csrc/dpmm_model.cpp) holding K clusters with given packed statistics, over the test worker."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_worker import FakeWorker
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3.0, np.eye(D))
    wk = FakeWorker(0, D, 0)
    s = host.DPMMSampler(wk, prior, alpha, 10000, seed=7, burnout=burnout, nthreads=2)
    s.f32_quirk = f32_quirk
    s._configure()
    stride = 1 + D + D * (D + 1) // 2
    packed = np.zeros((2 * K, stride))
    il = np.tril_indices(D)
    stats = []
    for k in range(K):
        c = rng.normal(size=D) * 6
        for side in range(2):
            n = int(rng.integers(30, 300))
            x = rng.normal(size=(n, D)) * (0.5 + rng.random()) + c + (0.8 if side else -0.8)
            packed[2 * k + side, 0] = n
            packed[2 * k + side, 1:1 + D] = x.sum(0)
            packed[2 * k + side, 1 + D:] = (x.T @ x)[il]
            stats.append((n, x.sum(0), x.T @ x))
    s.model.set("K", K)
    s.model.set("packed", packed)
    return s, prior, stats


def _oracle_L(prior, n, sm, S, f32_quirk=False):
    pr = (prior.kappa, prior.m, prior.nu, prior.psi)
    return orc.niw_log_marginal(pr, orc.niw_calc_posterior(*pr, n, sm, S), n, prior.dim, f32_quirk=f32_quirk)


@pytest.mark.parametrize("f32_quirk", [False, True])
def test_engine_split_and_merge_log_hastings_ratios_vs_oracle(host, f32_quirk):
    """should_split_local! (local_clusters_actions.jl:336-339) and should_merge! (shared_actions.jl:28-30) as the engine evaluates
    them on fixed statistics == orc.split_log_hr / orc.merge_log_hr on the oracle's posteriors and marginals."""
    D, K = 4, 5
    s, prior, stats = _engine_with_stats(host, D, K, np.random.default_rng(2), f32_quirk=f32_quirk)
    alpha = s.alpha
    s.model.set("splittable", np.ones(K, np.uint8))
    got_split = s.model.debug_split_log_hr()
    got_merge = s.model.debug_merge_log_hr()
    Lc, Nc = [], []
    for k in range(K):
        (nl, sl, Sl), (nr, sr, Sr) = stats[2 * k], stats[2 * k + 1]
        L_l, L_r = _oracle_L(prior, nl, sl, Sl, f32_quirk), _oracle_L(prior, nr, sr, Sr, f32_quirk)
        L_c = _oracle_L(prior, nl + nr, sl + sr, Sl + Sr, f32_quirk)
        Lc.append(L_c); Nc.append(nl + nr)
        want = orc.split_log_hr(alpha, nl, L_l, nr, L_r, nl + nr, L_c)
        assert abs(got_split[k] - want) <= 1e-9 * max(1.0, abs(want)), (k, got_split[k], want)
    # the engine's cached marginals are the oracle's
    np.testing.assert_allclose(s.log_marginals[:, 0], Lc, rtol=1e-10)
    for i in range(K):
        for j in range(i + 1, K):
            (nli, sli, Sli), (nri, sri, Sri) = stats[2 * i], stats[2 * i + 1]
            (nlj, slj, Slj), (nrj, srj, Srj) = stats[2 * j], stats[2 * j + 1]
            L = _oracle_L(prior, Nc[i] + Nc[j], sli + sri + slj + srj, Sli + Sri + Slj + Srj, f32_quirk)
            want = orc.merge_log_hr(alpha, Nc[i], Lc[i], Nc[j], Lc[j], L)
            assert abs(got_merge[i, j] - want) <= 1e-9 * max(1.0, abs(want)), (i, j, got_merge[i, j], want)
    # eligibility: not splittable -> no ratio (NaN); a cluster with an empty sub-cluster is not split (local_clusters_actions.jl:323)
    s.model.set("splittable", np.array([1, 0, 1, 1, 1], np.uint8))
    assert np.isnan(s.model.debug_split_log_hr()[1]) and np.isnan(s.model.debug_merge_log_hr()[1, 2]) and np.isnan(s.model.debug_merge_log_hr()[0, 1])
    # log posterior (dp-parallel-sampling.jl:458-470)
    want_lp = orc.log_posterior(alpha, 10000, Nc, Lc)
    assert abs(s.log_posterior() - want_lp) <= 1e-9 * abs(want_lp)


def test_engine_burn_in_gate_matches_hand_computation(host):
    """sample_cluster_params' gate (shared_actions.jl:51-63): shift the Float32 history, append L_l + L_r, average over
    `burnout` entries with divisor burnout - 0.1, splittable when avg != -Inf and avg - last < 1e-2; never reset here."""
    D, K, b = 3, 4, 5
    s, prior, stats = _engine_with_stats(host, D, K, np.random.default_rng(9), burnout=b)
    L = s.log_marginals
    last = (L[:, 1] + L[:, 2]).astype(np.float32)
    H = np.full((K, b + 5), -np.inf, np.float32)
    H[0, :b] = last[0] + np.float32([-0.05, -0.04, -0.03, -0.02, 0.0])      # plateau: mean just below the newest value -> gate opens
    H[1, :b] = last[1] + np.float32([-900, -700, -500, -300, 0])            # still climbing: mean far BELOW the newest -> avg - last < 1e-2 -> opens too
    H[2, :b] = [-np.inf, -np.inf, last[2], last[2], last[2]]                # -Inf inside the window -> mean -Inf -> stays closed
    H[3, :b] = last[3] + np.float32([5e5, 4e5, 3e5, 2e5, 1e5])              # falling steeply: mean far ABOVE the newest -> avg - last > 1e-2 -> closed
    # (the divisor is burnout - 0.1, not burnout: avg = sum / 4.9 carries an extra 0.0204 * last, hence the large steps)
    s.model.set("hist", H)
    s.model.set("splittable", np.array([0, 0, 0, 1], np.uint8))             # cluster 3 was splittable before: the gate never closes it
    s.model.sample_clusters()
    got_h, got_s = s.hist, s.splittable
    want_s = [None] * K
    for k in range(K):
        h = H[k].copy()
        h[:b - 1] = h[1:b]
        h[b - 1] = last[k]
        now = float(np.sum(h[:b].astype(np.float64) * (1.0 / (b - 0.1))))
        want_s[k] = bool((k == 3) or (now != -np.inf and now - float(h[b - 1]) < 1e-2))
        assert np.array_equal(got_h[k, :b], h[:b]), k
    assert got_s.tolist() == want_s == [True, True, False, True]
    # and a falling history closes nothing but opens nothing either: same state, cluster 3 not splittable beforehand
    s.model.set("hist", H)
    s.model.set("splittable", np.zeros(K, np.uint8))
    s.model.sample_clusters()
    assert s.splittable.tolist() == [True, True, False, False]
    # lr_weights ~ Dirichlet(N_l + alpha/2, N_r + alpha/2) and the mixture weights ~ Dirichlet(N_1..N_K, alpha)[1:K]: moments over epochs
    lrs, ws = [], []
    for _ in range(400):
        s.model.sample_clusters()
        lrs.append(s.lr_weights); ws.append(s.weights)
    lrs, ws = np.array(lrs), np.array(ws)
    N = s.N
    a = N[:, 1:3] + s.alpha / 2
    np.testing.assert_allclose(lrs.mean(0), a / a.sum(1, keepdims=True), atol=4 * np.sqrt(0.25 / a.sum(1).min() / 400) + 1e-3)
    conc = np.concatenate([N[:, 0], [s.alpha]])
    np.testing.assert_allclose(ws.mean(0), (conc / conc.sum())[:K], atol=5e-3)
    assert np.all(np.abs(lrs.sum(2) - 1) < 1e-6) and np.all(ws.sum(1) < 1.0)


def test_numa_option_and_prewake_are_harmless(host):
    """DPMMH_OPT_NUMA_NODE binds the pool to a node's CPUs (a missing node or an empty intersection is ignored) and
    DPMMH_OPT_PREWAKE toggles the timed wake-up: neither may change a result."""
    from fake_worker import FakeWorker
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    x, y = host.generate_gaussian_data(400, 2, 3, 60.0, seed=3)[:2]
    x = x.astype(np.float32)
    hyper = host.niw_hyperparams(1.0, np.zeros(2), 5, np.eye(2))
    labels = []
    for node, pre in ((-1, 1), (0, 0), (9999, 1)):
        wk = FakeWorker(0, 2, x.shape[1]); wk.upload_points(x.T.copy())
        s = host.DPMMSampler(wk, hyper, 10.0, x.shape[1], seed=5, burnout=3, nthreads=3)
        s._configure()
        s.model.set_option(engine.OPT_NUMA_NODE, node)
        s.model.set_option(engine.OPT_PREWAKE, pre)
        s.init_first_clusters(2)
        for _ in range(6):
            s.group_step(False, False)
        labels.append(wk.get_labels())
    for l in labels[1:]:
        assert np.array_equal(l[0], labels[0][0]) and np.array_equal(l[1], labels[0][1])
