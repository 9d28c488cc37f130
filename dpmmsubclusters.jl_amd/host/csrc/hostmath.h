// hostmath.h -- per-distribution dense maths of the master side (shared by the batch entry points in dpmm_host.cpp and
// the sweep engine in dpmm_model.cpp).  Reference functions restated (paths relative to the reference checkout):
//   calc_posterior           src/priors/niw.jl:20-31
//   sample_distribution      src/priors/niw.jl:34-40          (Sigma ~ InvWishart(nu, nu psi), mu ~ N(m, Sigma/kappa))
//   log_marginal_likelihood  src/priors/niw.jl:53-62, src/priors/multinomial_prior.jl:34-39
//   log_multivariate_gamma   src/utils.jl:66-72
//   sample_distribution      src/priors/multinomial_prior.jl:23-25 (log of a Dirichlet draw)
//
// Sampling without ever forming Sigma: with Psi = nu psi = U U' (U upper triangular, "reverse" Cholesky) and A
// lower-triangular Bartlett (A_ii^2 ~ chi2(nu - i), A_ij ~ N(0,1), i > j),
//   Sigma^-1 = W = (U^-T A)(U^-T A)' ~ Wishart(nu, Psi^-1),   R := A' U^-1  (upper),  W = R'R
// so the GPU's factor R comes out of one triangular solve, logdet Sigma = -2 sum log R_ii, and
// mu = m + R^-1 xi / sqrt(kappa).
#pragma once
#include <algorithm>

#include "hostlib.h"
#include "dense.h"

namespace dpmmh {

inline std::vector<double> &dense_scratch(size_t n) {
    static thread_local std::vector<double> buf;
    if (buf.size() < n) buf.resize(n);
    return buf;
}
// Psi (row-major, symmetric, D x D; the caller's matrix survives) = U U', U upper triangular (row-major, zeros below the
// diagonal).  Returns false when Psi is not positive definite.  Batch / compatibility form of chol_ltl (dense.h): U = L'.
inline bool reverse_cholesky(const double *P, int D, double *U) {
    const size_t DD = (size_t)D * D;
    std::vector<double> &w = dense_scratch(DD);
    memcpy(w.data(), P, sizeof(double) * DD);
    const double ld = chol_ltl(w.data(), D, nullptr);
    memset(U, 0, sizeof(double) * DD);
    if (!(ld == ld)) return false;
    for (int j = 0; j < D; ++j)
        for (int i = 0; i <= j; ++i) U[(size_t)i * D + j] = w[(size_t)j * D + i];
    return true;
}

// log det of a symmetric positive definite matrix (row-major, D x D, LOWER triangle read; destroyed); NaN when it is not
// positive definite.
inline double logdet_spd_inplace(double *P, int D) { return chol_ltl(P, D, nullptr); }

// priors/niw.jl:20-31 for one statistic set; psi_out symmetric.  N == 0 -> prior.
inline void niw_posterior_one(int D, double k0, double v0, const double *m0, const double *psi0, double N, const double *sum,
                              const double *S, double *kap, double *nu, double *m, double *psi) {
    if (N == 0.0) {
        *kap = k0; *nu = v0;
        memcpy(m, m0, sizeof(double) * D);
        memcpy(psi, psi0, sizeof(double) * (size_t)D * D);
        return;
    }
    const double k1 = k0 + N, v1 = v0 + N;
    *kap = k1; *nu = v1;
    for (int a = 0; a < D; ++a) m[a] = (m0[a] * k0 + sum[a]) / k1;
    for (int a = 0; a < D; ++a)
        for (int b = 0; b <= a; ++b) {
            const double sab = 0.5 * (S[(size_t)a * D + b] + S[(size_t)b * D + a]);
            const double pab = 0.5 * (psi0[(size_t)a * D + b] + psi0[(size_t)b * D + a]);
            const double v = (v0 * pab + k0 * m0[a] * m0[b] - k1 * m[a] * m[b] + sab) / v1;
            psi[(size_t)a * D + b] = v;
            psi[(size_t)b * D + a] = v;
        }
}

// Same posterior, from PACKED statistics rows {N, sum[D], lower triangle of S} (include/dpmm_hip.h): the statistic set is
// cl * (row l) + cr * (row r) (cluster = left + right: cl = cr = 1).  Writes kappa, nu, m and the lower triangle of the SCALE
// matrix P = nu' psi' that the factorisation consumes -- the full S is never materialised.  psi_lo: the prior's psi, symmetrised
// (0.5 (psi + psi')) and packed like the statistics (row a: columns 0..a at offset a (a+1)/2), so that every stream of the inner
// loop is contiguous (a column walk over the full psi cost more than the factorisation at D = 256).
inline void pack_sym_lower(int D, const double *psi, double *psi_lo) {
    for (int a = 0; a < D; ++a)
        for (int b = 0; b <= a; ++b) psi_lo[(size_t)a * (a + 1) / 2 + b] = 0.5 * (psi[(size_t)a * D + b] + psi[(size_t)b * D + a]);
}
inline double niw_posterior_packed(int D, double k0, double v0, const double *m0, const double *psi_lo, const double *l,
                                   const double *r, double cl, double cr, double *kap, double *nu, double *m, double *P) {
    const double N = cl * l[0] + cr * r[0];
    if (N == 0.0) {
        *kap = k0; *nu = v0;
        memcpy(m, m0, sizeof(double) * D);
        for (int a = 0; a < D; ++a)
            for (int b = 0; b <= a; ++b) P[(size_t)a * D + b] = psi_lo[(size_t)a * (a + 1) / 2 + b] * v0;
        return N;
    }
    const double k1 = k0 + N, v1 = v0 + N;
    *kap = k1; *nu = v1;
    const double *sl = l + 1, *sr = r + 1, *tl = l + 1 + D, *tr = r + 1 + D;
    for (int a = 0; a < D; ++a) m[a] = (m0[a] * k0 + (cl * sl[a] + cr * sr[a])) / k1;
    for (int a = 0; a < D; ++a) {
        const size_t t0 = (size_t)a * (a + 1) / 2;
        const double *ta = tl + t0, *tb = tr + t0, *pa = psi_lo + t0;
        double *Pa = P + (size_t)a * D;
        const double km0a = k0 * m0[a], kma = k1 * m[a];
#pragma omp simd
        for (int b = 0; b <= a; ++b) {
            const double sab = cl * ta[b] + cr * tb[b];
            Pa[b] = ((v0 * pa[b] + km0a * m0[b] - kma * m[b] + sab) / v1) * v1;   // psi' then nu' psi' (niw.jl:29,35); LOWER triangle
        }
    }
    return N;
}

// utils.jl:66-72.  f32_quirk: the reference's accumulator is a Float32 local.
inline double log_multivariate_gamma(double x, int D, bool f32_quirk) {
    int sg;
    if (f32_quirk) {
        float res = (float)((double)D * (D - 1) / 4.0 * log(M_PI));
        for (int d = 1; d <= D; ++d) res = (float)((double)res + lgamma_r(x + (1 - d) / 2.0, &sg));
        return (double)res;
    }
    double res = (double)D * (D - 1) / 4.0 * log(M_PI);
    for (int d = 1; d <= D; ++d) res += lgamma_r(x + (1 - d) / 2.0, &sg);
    return res;
}

// priors/niw.jl:53-62 from scalars.  lmg0 = log_multivariate_gamma(nu0 / 2) (constant of the prior).
inline double niw_log_marginal(int D, double k0, double v0, double logdet_psi0, double lmg0, double k1, double v1,
                               double logdet_psi1, double N, bool f32_quirk) {
    return -N * D * 0.5 * log(M_PI) + log_multivariate_gamma(v1 / 2.0, D, f32_quirk) - lmg0 +
           (v0 / 2.0) * (D * log(v0) + logdet_psi0) - (v1 / 2.0) * (D * log(v1) + logdet_psi1) + (D / 2.0) * log(k0 / k1);
}

// the same with log Gamma_D(nu' / 2) supplied (the device master returns it with the other scalars)
inline double niw_log_marginal_lmg(int D, double k0, double v0, double logdet_psi0, double lmg0, double k1, double v1,
                                   double logdet_psi1, double N, double lmg1) {
    return -N * D * 0.5 * log(M_PI) + lmg1 - lmg0 +
           (v0 / 2.0) * (D * log(v0) + logdet_psi0) - (v1 / 2.0) * (D * log(v1) + logdet_psi1) + (D / 2.0) * log(k0 / k1);
}

// Marsaglia-Tsang Gamma(a, 1), a >= 1, with the FIRST trial's (normal, uniform) pair supplied by the caller (it does not depend on the
// shape, so it can be generated ahead of time); later trials (~2 % of the draws) come from `retry`.
inline double gamma_first_trial(double a, double x, double u, Philox &retry) {      // shape a >= 1
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double v = 1.0 + c * x;
        if (v > 0.0) {
            v = v * v * v;
            if (u < 1.0 - 0.0331 * x * x * x * x) return d * v;
            if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return d * v;
        }
        x = retry.normal(); u = retry.uniform();
    }
}

// One draw (mu, R, logdet Sigma) from a prepared posterior (kappa, nu, m, L with nu psi = L' L, L lower triangular = U').
// `id`/`epoch` key the random streams (normals: stream 16 -- identical whether pre-generated or not; chi-squares: stream 18).
// An / xi_in: optional pre-generated standard normals (strictly-lower Bartlett entries row-major [D][D], and xi [D]).
// scratch: D^2 + 2 D doubles.  mu_out [D]; R_out Float32: [D*D] full (upper, zeros below) or, r_packed, the packed upper
// triangle [D(D+1)/2] of the worker's parameter staging (row r: columns r..D-1 at offset r D - r (r-1)/2).
inline void niw_draw_one(int D, double kappa, double nu, const double *m, const double *Li, uint64_t seed, uint32_t id,
                         uint32_t epoch, const double *An, const double *xi_in, double *scratch, float *mu_out, float *R_out,
                         float *logdet_sigma, bool r_packed = false, double *An_inplace = nullptr, const double *chi_x = nullptr,
                         const double *chi_u = nullptr) {
    const size_t DD = (size_t)D * D;
    // An_inplace: a pre-generated noise block the caller gives up -- strictly-lower normals, ZEROS above the diagonal (the solve keeps
    // them zero): the system is solved where the noise lies, no copy of the D x D block
    double *Y = An_inplace ? An_inplace : scratch, *xi = scratch + DD, *v = scratch + DD + D;
    // chi-squares: stream 18 holds the first-trial pairs (chi_x / chi_u when pre-generated, the same values inline otherwise), 24 the rest
    Philox rng(seed, id, epoch, 16u), chi_first(seed, id, epoch, 18u), chi_retry(seed, id, epoch, 24u);
    // Bartlett factor A, lower triangular (chi on the diagonal, standard normals below): the right-hand side of  L Y = A
    for (int r = 0; r < D; ++r) {
        double *yr = Y + (size_t)r * D;
        if (!An_inplace) {
            if (An) memcpy(yr, An + (size_t)r * D, sizeof(double) * r);
            else for (int c = 0; c < r; ++c) yr[c] = rng.normal();
            memset(yr + r + 1, 0, sizeof(double) * (D - 1 - r));
        }
        {
            const double a = 0.5 * (nu - r);
            const double x = chi_x ? chi_x[r] : chi_first.normal();
            const double u = chi_u ? chi_u[r] : chi_first.uniform();
            const double g = a >= 1.0 ? gamma_first_trial(a, x, u, chi_retry)
                                      : gamma_first_trial(a + 1.0, x, u, chi_retry) * pow(chi_retry.uniform(), 1.0 / a);
            yr[r] = sqrt(2.0 * g);
        }
    }
    solve_lower_left(Y, Li, D);            // Y = L^-1 A;  R = Y' = A' U^-1 (upper)
    double ld = 0.0;
    for (int j = 0; j < D; ++j) ld += log(Y[(size_t)j * D + j]);
    *logdet_sigma = (float)(-2.0 * ld);
    // mu = m + R^-1 xi / sqrt(kappa):  R v = xi by back substitution, in the column (axpy) form: column c of R = row c of Y
    for (int d = 0; d < D; ++d) xi[d] = xi_in ? xi_in[d] : rng.normal();
    for (int c = D - 1; c >= 0; --c) {
        const double *yc = Y + (size_t)c * D;
        const double vc = xi[c] / yc[c];
        v[c] = vc;
#pragma omp simd
        for (int r = 0; r < c; ++r) xi[r] -= vc * yc[r];
    }
    const double isk = 1.0 / sqrt(kappa);
    for (int d = 0; d < D; ++d) mu_out[d] = (float)(m[d] + v[d] * isk);
    // R = Y' as Float32: row r of R = column r of Y from the diagonal down, 8 x 8 tiles (R_out packed or full, see r_packed)
    for (int r0 = 0; r0 < D; r0 += 8)
        for (int c0 = r0; c0 < D; c0 += 8) {
            const int r1 = std::min(D, r0 + 8), c1 = std::min(D, c0 + 8);
            for (int r = r0; r < r1; ++r) {
                float *dst = r_packed ? R_out + (size_t)r * D - (size_t)r * (r - 1) / 2 - r : R_out + (size_t)r * D;     // dst[c], c >= r
                for (int c = std::max(c0, r); c < c1; ++c) dst[c] = (float)Y[(size_t)c * D + r];
            }
        }
    if (!r_packed)
        for (int r = 1; r < D; ++r) memset(R_out + (size_t)r * D, 0, sizeof(float) * r);
}
inline size_t niw_draw_scratch_doubles(int D) { return (size_t)D * D + 2 * (size_t)D; }

// Standard-normal noise of one draw (see niw_draw_one): depends on (seed, epoch, id) only.
inline void niw_noise_one(int D, uint64_t seed, uint32_t id, uint32_t epoch, double *A, double *xi, double *chi_x = nullptr, double *chi_u = nullptr) {
    Philox rng(seed, id, epoch, 16u);
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < r; ++c) A[(size_t)r * D + c] = rng.normal();
    for (int d = 0; d < D; ++d) xi[d] = rng.normal();
    if (chi_x && chi_u) {
        Philox first(seed, id, epoch, 18u);
        for (int r = 0; r < D; ++r) { chi_x[r] = first.normal(); chi_u[r] = first.uniform(); }
    }
}

// log of a Dirichlet(alpha) draw (priors/multinomial_prior.jl:23-25): logp[d] = log(g_d / sum g), g_d ~ Gamma(alpha_d).
// Works with log-gammas so that tiny shapes do not underflow: log g = log Gamma(a+1) draw + log(u)/a.   lg: D doubles.
// Randomness: stream 19 supplies ONE (normal, uniform) pair per component -- the first Marsaglia-Tsang trial, which is accepted
// ~98 % of the time; it does not depend on the shapes, so dirichlet_noise_one can produce it ahead of time (px / pu, while the GPU
// sweeps) and the draw is identical whether it was pre-generated or not.  Retries and the extra uniform of shapes < 1: stream 17.
inline void dirichlet_noise_one(int D, uint64_t seed, uint32_t id, uint32_t epoch, double *px, double *pu) {
    Philox rng(seed, id, epoch, 19u);
    for (int d = 0; d < D; ++d) { px[d] = rng.normal(); pu[d] = rng.uniform(); }
}
inline void dirichlet_log_one(int D, const float *al, uint64_t seed, uint32_t id, uint32_t epoch, double *lg, float *logp,
                              const double *px = nullptr, const double *pu = nullptr) {
    Philox first(seed, id, epoch, 19u), rng(seed, id, epoch, 17u);
    double mx = -INFINITY;
    for (int d = 0; d < D; ++d) {
        const double a = (double)al[d];
        const double x = px ? px[d] : first.normal();
        const double u = pu ? pu[d] : first.uniform();
        double l;
        if (a < 1.0) l = log(gamma_first_trial(a + 1.0, x, u, rng)) + log(rng.uniform()) / a;
        else l = log(gamma_first_trial(a, x, u, rng));
        lg[d] = l;
        if (l > mx) mx = l;
    }
    double s = 0.0;
    for (int d = 0; d < D; ++d) s += exp(lg[d] - mx);
    const double lse = mx + log(s);
    for (int d = 0; d < D; ++d) logp[d] = (float)(lg[d] - lse);
}

// lgamma at small positive INTEGERS from a table of libm's own values (built once, identical bits): count data with an integer
// Dirichlet prior evaluates lgamma only there, 2 D times per distribution and per merge pair -- the table turns the master's
// Multinomial log-marginals from libm-bound into L1-bound.  Anything else goes to lgamma_r.
class LgammaTable {
  public:
    static constexpr int kSize = 1 << 16;
    static const LgammaTable &get() { static LgammaTable t; return t; }
    inline double operator()(float a) const {
        const int i = (int)a;
        if ((float)i == a && i >= 1 && i < kSize) return tab_[i];
        int sg;
        return lgamma_r((double)a, &sg);
    }
  private:
    LgammaTable() : tab_(kSize) {
        int sg;
        tab_[0] = INFINITY;
        for (int i = 1; i < kSize; ++i) tab_[i] = lgamma_r((double)i, &sg);
    }
    std::vector<double> tab_;
};

// priors/multinomial_prior.jl:34-39 in Float64: lgamma(sum a0) - lgamma(sum a1) + sum (lgamma(a1_d) - lgamma(a0_d)).
inline double mult_log_marginal(int D, const float *a0, const float *a1) {
    int sg;
    const LgammaTable &lg = LgammaTable::get();
    double s0 = 0.0, s1 = 0.0, acc = 0.0;
    for (int d = 0; d < D; ++d) {
        s0 += (double)a0[d]; s1 += (double)a1[d];
        acc += lg(a1[d]) - lg(a0[d]);
    }
    return lgamma_r(s0, &sg) - lgamma_r(s1, &sg) + acc;
}

// A two-category Dirichlet / a K+1-category Dirichlet through Gamma draws (Distributions.jl's Dirichlet sampler is
// un-vendored; the stream is this library's own): stream 20 + caller-chosen id / epoch.
inline void dirichlet2(double a, double b, uint64_t seed, uint32_t id, uint32_t epoch, uint32_t stream, float out[2]) {
    Philox rng(seed, id, epoch, stream);
    const double g0 = rng.gamma(a), g1 = rng.gamma(b);
    const double s = g0 + g1;
    out[0] = (float)(g0 / s); out[1] = (float)(g1 / s);
}

}  // namespace dpmmh
