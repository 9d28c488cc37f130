// dpmm_host.cpp -- native host-side maths of the sampler (libdpmmhost.so, plain C++/OpenMP).
//
// north_star keeps "posterior cluster-parameter draws and split/merge Metropolis steps" on the
// host; at K ~ 32, D = 64..256 these are 3K dense D x D factorisations per sweep and would
// dominate a ~1.5 ms GPU sweep if left to an interpreter, so they are native and threaded over
// clusters.  Reference functions restated (paths relative to the reference checkout):
//   calc_posterior          src/priors/niw.jl:20-31
//   sample_distribution     src/priors/niw.jl:34-40   (Sigma ~ InvWishart(nu, nu psi), mu ~ N(m, Sigma/kappa))
//   log_marginal_likelihood src/priors/niw.jl:53-62   (only the logdet(psi) terms are computed here)
//   sample_distribution     src/priors/multinomial_prior.jl:23-25 (log of a Dirichlet draw)
//
// Sampling without ever forming Sigma: with Psi = nu psi = U U' (U upper triangular, "reverse"
// Cholesky) and A lower-triangular Bartlett (A_ii^2 ~ chi2(nu - i), A_ij ~ N(0,1), i > j),
//   Sigma^-1 = W = (U^-T A)(U^-T A)' ~ Wishart(nu, Psi^-1),   R := A' U^-1  (upper),  W = R'R
// so the GPU's factor R comes out of one triangular solve, logdet Sigma = -2 sum log R_ii, and
// mu = m + R^-1 xi / sqrt(kappa).  Randomness: Philox4x32-10 keyed by (seed; draw id, epoch),
// so results do not depend on thread count or batch composition (all ranks of a multi-GPU run
// draw identical parameters from identical all-reduced statistics).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "hostlib.h"
#include "hostmath.h"

#define HAPI extern "C" __attribute__((visibility("default")))

using dpmmh::Pool;
using dpmmh::Philox;
using dpmmh::reverse_cholesky;
using dpmmh::niw_posterior_one;

// Batch posterior + factorisation.  Outputs: kappa[n], nu[n], m[n][D], psi[n][D*D] (may be NULL),
// U[n][D*D] with nu*psi = U U' (may be NULL), logdet_psi[n] (NaN when psi is not positive definite).
HAPI int dpmmh_niw_posterior(int n, int D, double kappa0, double nu0, const double *m0, const double *psi0,
                             const double *N, const double *sum, const double *S, double *kappa, double *nu, double *m,
                             double *psi, double *U, double *logdet_psi, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    {
        std::vector<std::vector<double>> scratch(nthreads, std::vector<double>(3 * (size_t)D * D));
        Pool::get().run(n, nthreads, [&](int i, int slot) {
            double *P_ = scratch[slot].data(), *Ul_ = P_ + (size_t)D * D, *pl_ = Ul_ + (size_t)D * D;
            struct { double *p; double *data() { return p; } double &operator[](size_t e) { return p[e]; } } P{P_}, Ul{Ul_}, pl{pl_};
            double *pp = psi ? psi + (size_t)i * D * D : pl.data();
            niw_posterior_one(D, kappa0, nu0, m0, psi0, N[i], sum + (size_t)i * D, S + (size_t)i * D * D, &kappa[i], &nu[i],
                              m + (size_t)i * D, pp);
            for (size_t e = 0; e < (size_t)D * D; ++e) P[e] = pp[e] * nu[i];
            double *Uo = U ? U + (size_t)i * D * D : Ul.data();
            if (reverse_cholesky(P.data(), D, Uo)) {
                double ld = 0.0;
                for (int d = 0; d < D; ++d) ld += log(Uo[(size_t)d * D + d]);
                logdet_psi[i] = 2.0 * ld - D * log(nu[i]);
            } else {
                logdet_psi[i] = NAN;
            }
        });
    }
    return 0;
}

// logdet(psi') of the posterior for the MERGED statistics of cluster pairs (shared_actions.jl:21-38 needs
// log_marginal_likelihood of the pooled cluster).  pairs[2p], pairs[2p+1] index rows of N/sum/S.
HAPI int dpmmh_niw_logdet_pairs(int npairs, const int32_t *pairs, int D, double kappa0, double nu0, const double *m0,
                                const double *psi0, const double *N, const double *sum, const double *S,
                                double *logdet_psi, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    {
        const size_t DD = (size_t)D * D;
        std::vector<std::vector<double>> scratch(nthreads, std::vector<double>(3 * DD + 2 * (size_t)D));
        Pool::get().run(npairs, nthreads, [&](int p, int slot) {
            struct V { double *p; double *data() { return p; } double &operator[](size_t e) { return p[e]; } };
            double *base = scratch[slot].data();
            V Sm{base}, P{base + DD}, Ul{base + 2 * DD}, sm{base + 3 * DD}, mm{base + 3 * DD + D};
            const int a = pairs[2 * p], b = pairs[2 * p + 1];
            const double *Sa = S + (size_t)a * D * D, *Sb = S + (size_t)b * D * D;
            for (size_t e = 0; e < (size_t)D * D; ++e) Sm[e] = Sa[e] + Sb[e];
            for (int d = 0; d < D; ++d) sm[d] = sum[(size_t)a * D + d] + sum[(size_t)b * D + d];
            double kap, nu;
            niw_posterior_one(D, kappa0, nu0, m0, psi0, N[a] + N[b], sm.data(), Sm.data(), &kap, &nu, mm.data(), P.data());
            for (size_t e = 0; e < (size_t)D * D; ++e) P[e] *= nu;
            if (reverse_cholesky(P.data(), D, Ul.data())) {
                double ld = 0.0;
                for (int d = 0; d < D; ++d) ld += log(Ul[(size_t)d * D + d]);
                logdet_psi[p] = 2.0 * ld - D * log(nu);
            } else {
                logdet_psi[p] = NAN;
            }
        });
    }
    return 0;
}

// Draw (mu, R, logdet Sigma) for n prepared posteriors.  ids[i] keys the random stream of draw i.
// want_sigma: also return Sigma (Float32 [n][D*D]) -- only needed for the user-facing result of fit().
// Standard-normal noise of the draws (strictly-lower Bartlett entries, row-major [n][D*D], and xi [n][D]).
// It depends on (seed, epoch, id) only -- not on the statistics -- so the sampler generates it while the GPU
// sweeps and hands it to dpmmh_niw_sample_noise afterwards.
HAPI int dpmmh_niw_noise(int n, int D, uint64_t seed, uint32_t epoch, const int32_t *ids, double *A_noise, double *xi,
                         int nthreads) {
    if (nthreads < 1) nthreads = 1;
    Pool::get().run(n, nthreads, [&](int i, int) {
        dpmmh::niw_noise_one(D, seed, (uint32_t)ids[i], epoch, A_noise + (size_t)i * D * D, xi + (size_t)i * D);
    });
    return 0;
}

static int niw_sample_impl(int n, int D, const double *kappa, const double *nu, const double *m, const double *U,
                           uint64_t seed, uint32_t epoch, const int32_t *ids, const double *A_noise, const double *xi_in,
                           float *mu, float *R, float *logdet_sigma, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    const size_t DD = (size_t)D * D;
    std::vector<std::vector<double>> scratch(nthreads, std::vector<double>(dpmmh::niw_draw_scratch_doubles(D) + DD));
    Pool::get().run(n, nthreads, [&](int i, int slot) {
        double *Lt = scratch[slot].data() + dpmmh::niw_draw_scratch_doubles(D);      // the batch API hands U (upper): L = U'
        const double *Ui = U + (size_t)i * DD;
        for (int r = 0; r < D; ++r)
            for (int c = 0; c < D; ++c) Lt[(size_t)r * D + c] = c <= r ? Ui[(size_t)c * D + r] : 0.0;
        dpmmh::niw_draw_one(D, kappa[i], nu[i], m + (size_t)i * D, Lt, seed, (uint32_t)ids[i], epoch,
                            A_noise ? A_noise + (size_t)i * DD : nullptr, xi_in ? xi_in + (size_t)i * D : nullptr,
                            scratch[slot].data(), mu + (size_t)i * D, R + (size_t)i * DD, &logdet_sigma[i]);
    });
    return 0;
}

HAPI int dpmmh_niw_sample_noise(int n, int D, const double *kappa, const double *nu, const double *m, const double *U,
                                uint64_t seed, uint32_t epoch, const int32_t *ids, const double *A_noise, const double *xi,
                                float *mu, float *R, float *logdet_sigma, int nthreads) {
    return niw_sample_impl(n, D, kappa, nu, m, U, seed, epoch, ids, A_noise, xi, mu, R, logdet_sigma, nthreads);
}

HAPI int dpmmh_niw_sample(int n, int D, const double *kappa, const double *nu, const double *m, const double *U,
                          uint64_t seed, uint32_t epoch, const int32_t *ids, float *mu, float *R, float *logdet_sigma,
                          int nthreads) {
    return niw_sample_impl(n, D, kappa, nu, m, U, seed, epoch, ids, nullptr, nullptr, mu, R, logdet_sigma, nthreads);
}

// Sigma^-1 = R'R and Sigma = (R'R)^-1 in Float64 from the Float32 factor (user-facing mv_gaussian
// fields of the fit() result, distributions/mv_gaussian.jl:12-18; never used on the hot path).
HAPI int dpmmh_niw_expand(int n, int D, const float *R, double *inv_sigma, double *sigma, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    Pool::get().run(n, nthreads, [&](int i, int) {
        const float *Ri = R + (size_t)i * D * D;
        double *W = inv_sigma + (size_t)i * D * D;
        for (int a = 0; a < D; ++a)
            for (int b = 0; b <= a; ++b) {
                double s = 0.0;
                for (int k = 0; k <= b; ++k) s += (double)Ri[(size_t)k * D + a] * (double)Ri[(size_t)k * D + b];
                W[(size_t)a * D + b] = W[(size_t)b * D + a] = s;
            }
        if (sigma) {
            // Sigma = R^-1 R^-T : invert the upper-triangular factor, then multiply
            std::vector<double> Ri64((size_t)D * D, 0.0), Inv((size_t)D * D, 0.0);
            for (size_t e = 0; e < (size_t)D * D; ++e) Ri64[e] = Ri[e];
            for (int c = 0; c < D; ++c) {  // column c of R^-1
                for (int r = c; r >= 0; --r) {
                    double s = (r == c) ? 1.0 : 0.0;
                    for (int k = r + 1; k <= c; ++k) s -= Ri64[(size_t)r * D + k] * Inv[(size_t)k * D + c];
                    Inv[(size_t)r * D + c] = s / Ri64[(size_t)r * D + r];
                }
            }
            double *Sg = sigma + (size_t)i * D * D;
            for (int a = 0; a < D; ++a)
                for (int b = 0; b <= a; ++b) {
                    double s = 0.0;
                    for (int k = a; k < D; ++k) s += Inv[(size_t)a * D + k] * Inv[(size_t)b * D + k];
                    Sg[(size_t)a * D + b] = Sg[(size_t)b * D + a] = s;
                }
        }
    });
    return 0;
}

// log of Dirichlet(alpha) draws (priors/multinomial_prior.jl:23-25): logp[i][d] = log(g_d / sum g), g_d ~ Gamma(alpha_d)
HAPI int dpmmh_dirichlet_log(int n, int D, const float *alpha, uint64_t seed, uint32_t epoch, const int32_t *ids,
                             float *logp, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    std::vector<std::vector<double>> scratch(nthreads, std::vector<double>(D));
    Pool::get().run(n, nthreads, [&](int i, int slot) {
        dpmmh::dirichlet_log_one(D, alpha + (size_t)i * D, seed, (uint32_t)ids[i], epoch, scratch[slot].data(), logp + (size_t)i * D);
    });
    return 0;
}

// log_marginal_likelihood of n prepared posteriors (priors/niw.jl:53-62); f32_quirk reproduces utils.jl:66-72.
HAPI int dpmmh_niw_log_marginal(int n, int D, double kappa0, double nu0, double logdet_psi0, const double *kappa,
                                const double *nu, const double *logdet_psi, const double *N, int f32_quirk, double *out) {
    const double lmg0 = dpmmh::log_multivariate_gamma(nu0 / 2.0, D, f32_quirk != 0);
    for (int i = 0; i < n; ++i)
        out[i] = dpmmh::niw_log_marginal(D, kappa0, nu0, logdet_psi0, lmg0, kappa[i], nu[i], logdet_psi[i], N[i], f32_quirk != 0);
    return 0;
}

HAPI void dpmmh_set_spin_us(int us) { Pool::get().set_spin_us(us); }

HAPI int dpmmh_max_threads(void) { return (int)std::thread::hardware_concurrency(); }

// self-test of the thread pool (used by tests): returns the number of distinct slots that executed items
HAPI int dpmmh_pool_selftest(int n, int nthreads, int spin) {
    std::vector<int> used(nthreads > 0 ? nthreads : 1, 0);
    std::vector<double> sink(n > 0 ? n : 1);
    Pool::get().run(n, nthreads, [&](int i, int slot) {
        double s = 0;
        for (int k = 0; k < spin; ++k) s += sin(k * 1e-3 + i);
        sink[i] = s;
        used[slot] += 1;
    });
    int c = 0;
    for (int u : used) c += (u > 0);
    if (spin < 0) return 0;
    if (getenv("DPMM_POOL_DEBUG")) { for (int u : used) fprintf(stderr, "%d ", u); fprintf(stderr, "\n"); }
    return c;
}
