"""Prior plug-in surface (`distribution_hyper_params`) of the reference, batched.

The reference dispatches per-object functions on the abstract types of src/ds.jl:1-4
(contract: docs/src/priors.md:22-78).  Here a prior object answers the same questions for a
BATCH of statistic sets at once (struct-of-arrays over clusters x {cluster, left, right}),
because the host step of a sweep touches all 3K of them:

    reference function (per object)                          here (per batch)
    calc_posterior(prior, suff)                        ->    prior.posterior(N, sums, S)
    log_marginal_likelihood(prior, post, suff)         ->    prior.log_marginal(post, N)
    sample_distribution(post)                          ->    prior.sample(post, seed, epoch, ids)
    create_sufficient_statistics / aggregate_suff_stats ->   GPU (libdpmmhip) + elementwise sums
    log_likelihood!(r, x, dist)                        ->    GPU (libdpmmhip)

Cited reference code: src/priors/niw.jl, src/priors/multinomial_prior.jl,
src/distributions/mv_gaussian.jl, src/distributions/multinomial_dist.jl, src/utils.jl:66-72.
"""
from dataclasses import dataclass

import numpy as np
from scipy.special import gammaln

from . import native

PRIOR_NIW, PRIOR_MULT = 0, 1


class distribution_hyper_params:  # src/ds.jl:1
    kind = None


@dataclass
class mv_gaussian:  # src/distributions/mv_gaussian.jl:12-18
    mu: np.ndarray       # μ   (D,)  float32
    sigma: np.ndarray    # Σ   (D,D) float32
    inv_sigma: np.ndarray  # invΣ
    logdet_sigma: float  # logdetΣ
    inv_chol: np.ndarray  # upper-triangular R with invΣ = R'R (the reference's invChol = cholesky(invΣ).U)

    μ = property(lambda self: self.mu)
    Σ = property(lambda self: self.sigma)
    invΣ = property(lambda self: self.inv_sigma)
    logdetΣ = property(lambda self: self.logdet_sigma)


@dataclass
class multinomial_dist:  # src/distributions/multinomial_dist.jl:8-10
    alpha: np.ndarray    # log-probabilities (D,) float32

    α = property(lambda self: self.alpha)


class niw_hyperparams(distribution_hyper_params):
    """niw_hyperparams(κ, m, ν, ψ)  -- src/priors/niw.jl:6-11.  ψ is mean-like: the inverse-Wishart scale is ν·ψ."""
    kind = PRIOR_NIW

    def __init__(self, kappa, m, nu, psi):
        self.kappa = float(np.float32(kappa))   # κ, ν are Float32 in the reference
        self.nu = float(np.float32(nu))
        self.m = np.array(m, dtype=np.float64).ravel()
        self.psi = np.array(psi, dtype=np.float64).reshape(len(self.m), len(self.m))
        self.dim = len(self.m)
        self._logdet_psi = float(np.linalg.slogdet(self.psi)[1])

    κ = property(lambda self: self.kappa)
    ν = property(lambda self: self.nu)
    ψ = property(lambda self: self.psi)

    # ---- statistics layout: N (n,), sums (n,D), S (n,D,D)
    def empty_stats(self, n):
        D = self.dim
        return dict(N=np.zeros(n), sums=np.zeros((n, D)), S=np.zeros((n, D, D)))

    def posterior(self, N, sums, S, nthreads=None):
        """Batch calc_posterior (niw.jl:20-31) + factorisation ν'ψ' = U U'."""
        kap, nu, m, _, U, ld = native.niw_posterior(self.kappa, self.nu, self.m, self.psi, N, sums, S, nthreads=nthreads)
        return dict(kappa=kap, nu=nu, m=m, U=U, logdet_psi=ld)

    def empty_post(self, n):
        D = self.dim
        return dict(kappa=np.zeros(n), nu=np.zeros(n), m=np.zeros((n, D)), U=np.zeros((n, D, D)), logdet_psi=np.zeros(n))

    def log_marginal(self, post, N, f32_quirk=False):
        """niw.jl:53-62.  By default lnΓ_D is accumulated in Float64; `f32_quirk=True` reproduces the reference's
        log_multivariate_gamma (utils.jl:66-72), whose accumulator is a Float32 local -- the switch that makes a
        log-posterior trajectory comparable with the reference's own printed values."""
        D = self.dim
        N = np.asarray(N, float)
        v0, k0 = self.nu, self.kappa
        v1, k1 = post["nu"], post["kappa"]
        lmg = _lmvgamma_f32 if f32_quirk else _lmvgamma
        return (-N * D * 0.5 * np.log(np.pi) + lmg(v1 / 2, D) - lmg(np.float64(v0 / 2), D)
                + (v0 / 2) * (D * np.log(v0) + self._logdet_psi) - (v1 / 2) * (D * np.log(v1) + post["logdet_psi"])
                + (D / 2) * np.log(k0 / k1))

    def log_marginal_pairs(self, pairs, stats, nthreads=None, f32_quirk=False):
        """log_marginal_likelihood of the pooled statistics of cluster pairs (shared_actions.jl:22-27)."""
        N = stats["N"]
        ld = native.niw_logdet_pairs(pairs, self.kappa, self.nu, self.m, self.psi, N, stats["sums"], stats["S"], nthreads=nthreads)
        pairs = np.asarray(pairs).reshape(-1, 2)
        Np = N[pairs[:, 0]] + N[pairs[:, 1]]
        post = dict(nu=self.nu + Np, kappa=self.kappa + Np, logdet_psi=ld)
        return self.log_marginal(post, Np, f32_quirk=f32_quirk)

    def sample(self, post, seed, epoch, ids, nthreads=None, noise=None):
        """Batch sample_distribution (niw.jl:34-40): μ, R (invΣ = R'R), logdetΣ as Float32.
        `noise` = (A, xi) from `draw_noise` for the same (seed, epoch, ids[0..n)) -- identical result, less work."""
        if noise is not None and noise[0].shape[0] >= len(ids):
            mu, R, ld = native.niw_sample_noise(post["kappa"], post["nu"], post["m"], post["U"], seed, epoch, ids,
                                                noise[0], noise[1], nthreads=nthreads)
        else:
            mu, R, ld = native.niw_sample(post["kappa"], post["nu"], post["m"], post["U"], seed, epoch, ids, nthreads=nthreads)
        return dict(mu=mu, R=R, logdet=ld)

    def empty_params(self, n):
        D = self.dim
        return dict(mu=np.zeros((n, D), np.float32), R=np.zeros((n, D, D), np.float32), logdet=np.zeros(n, np.float32))

    def draw_noise(self, n, seed, epoch, nthreads=None):
        """Statistics-independent part of `sample` (standard normals); the sampler runs it while the GPU sweeps."""
        return native.niw_noise(n, self.dim, seed, epoch, np.arange(n), nthreads=nthreads)

    def upload(self, worker, params, lr_weights, weights):
        K = len(weights)
        D = self.dim
        worker.set_params_niw_chol(params["mu"].reshape(3 * K, D), params["R"].reshape(3 * K, D * D),
                                   params["logdet"].reshape(3 * K), lr_weights, weights)

    def distributions(self, params, rows):
        inv, sig = native.niw_expand(params["R"][rows])
        return [mv_gaussian(params["mu"][r].copy(), sig[i].astype(np.float32), inv[i].astype(np.float32),
                            float(params["logdet"][r]), params["R"][r].copy()) for i, r in enumerate(rows)]

    def predictive_table(self, worker, post, rows, weights, points=False):
        """posterior_predictive! (niw.jl:68-76): MvTDist(nu-D+1, m, ((kappa+1)/(kappa (nu-D+1))) nu psi) per cluster;
        returns parr[k][i] = logpdf + log w_k from the GPU (Student-t mode of the sweep kernel)."""
        D = self.dim
        kap, nu, m, U = post["kappa"][rows], post["nu"][rows], post["m"][rows], post["U"][rows]
        df = nu - D + 1
        c = (kap + 1) / (kap * df)
        Uinv = np.linalg.inv(U)                                # upper triangular: (nu psi)^-1 = Uinv' Uinv
        R = Uinv / np.sqrt(c)[:, None, None]                   # Sigma_t^-1 = R'R
        logdet = D * np.log(c) + 2 * np.log(np.einsum("kii->ki", U)).sum(1)
        if points:
            return worker.predict_table_niw(m, R.reshape(len(rows), -1), logdet, df, weights, points=True)
        return worker.predict_table_niw(m, R.reshape(len(rows), -1), logdet, df, weights)

    def posterior_hyperparams(self, post, row):
        """The reference's per-cluster posterior_hyperparams object (niw_hyperparams)."""
        U = post["U"][row]
        return niw_hyperparams(post["kappa"][row], post["m"][row], post["nu"][row], (U @ U.T) / post["nu"][row])


def _lmvgamma(x, D):
    x = np.asarray(x, float)
    d = np.arange(1, D + 1)
    return D * (D - 1) / 4 * np.log(np.pi) + gammaln(x[..., None] + (1 - d) / 2).sum(-1)


def _lmvgamma_f32(x, D):
    """log_multivariate_gamma exactly as utils.jl:66-72 evaluates it: `res::Float32`, every partial sum rounded to Float32."""
    x = np.atleast_1d(np.asarray(x, float))
    res = np.full(x.shape, np.float32(D * (D - 1) / 4 * np.log(np.pi)), np.float32)
    for d in range(1, D + 1):
        res = (res.astype(np.float64) + gammaln(x + (1 - d) / 2)).astype(np.float32)
    return res.astype(np.float64)


class multinomial_hyper(distribution_hyper_params):
    """multinomial_hyper(α)  -- src/priors/multinomial_prior.jl:6-8 (Dirichlet prior, Float32)."""
    kind = PRIOR_MULT

    def __init__(self, alpha):
        self.alpha = np.array(alpha, dtype=np.float32).ravel()
        self.dim = len(self.alpha)

    α = property(lambda self: self.alpha)

    def empty_stats(self, n):
        return dict(N=np.zeros(n), sums=np.zeros((n, self.dim)), S=None)

    def posterior(self, N, sums, S=None, nthreads=None):
        """multinomial_prior.jl:16-21: α' = α + Σx in Float32 (Σx is stored as Float32 by the reference)."""
        N = np.asarray(N, float)
        post = self.alpha[None, :] + np.asarray(sums, np.float64).astype(np.float32)
        post = np.where((N == 0)[:, None], self.alpha[None, :], post).astype(np.float32)
        return dict(alpha=post)

    def log_marginal(self, post, N, f32_quirk=False):
        """multinomial_prior.jl:34-39 (Float64 evaluation; `f32_quirk` has no counterpart for this prior)."""
        a = self.alpha.astype(np.float64)
        b = post["alpha"].astype(np.float64)
        return gammaln(a.sum()) - gammaln(b.sum(-1)) + (gammaln(b) - gammaln(a)).sum(-1)

    def log_marginal_pairs(self, pairs, stats, nthreads=None, f32_quirk=False):
        pairs = np.asarray(pairs).reshape(-1, 2)
        s = stats["sums"][pairs[:, 0]] + stats["sums"][pairs[:, 1]]
        Np = stats["N"][pairs[:, 0]] + stats["N"][pairs[:, 1]]
        return self.log_marginal(self.posterior(Np, s), Np)

    def sample(self, post, seed, epoch, ids, nthreads=None):
        """multinomial_prior.jl:23-25: log.(rand(Dirichlet(α')))"""
        return dict(logp=native.dirichlet_log(post["alpha"], seed, epoch, ids, nthreads=nthreads))

    def empty_params(self, n):
        return dict(logp=np.zeros((n, self.dim), np.float32))

    def upload(self, worker, params, lr_weights, weights):
        K = len(weights)
        worker.set_params_mult(params["logp"].reshape(3 * K, self.dim), lr_weights, weights)

    def distributions(self, params, rows):
        return [multinomial_dist(params["logp"][r].copy()) for r in rows]

    def predictive_table(self, worker, post, rows, weights, points=False):
        """posterior_predictive! (multinomial_prior.jl:45-48): log(alpha'/sum(alpha'))' x"""
        a = post["alpha"][rows].astype(np.float64)
        if points:
            return worker.predict_table_mult(np.log(a / a.sum(1, keepdims=True)), weights, points=True)
        return worker.predict_table_mult(np.log(a / a.sum(1, keepdims=True)), weights)

    def posterior_hyperparams(self, post, row):
        return multinomial_hyper(post["alpha"][row])
