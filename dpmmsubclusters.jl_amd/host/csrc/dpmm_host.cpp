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

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define HAPI extern "C" __attribute__((visibility("default")))

namespace {

// Passive thread pool.  OpenMP's idle workers spin-wait after every parallel region; in a container
// with a CFS CPU quota (the GPU box: 256 visible CPUs, 16 CPUs of quota) that burns the quota and
// the whole process gets throttled for ~50 ms every 100 ms.  Workers here sleep on a condition
// variable between jobs; items are handed out dynamically through an atomic counter.
class Pool {
  public:
    static Pool &get() { static Pool p; return p; }
    // run fn(item, slot) for item in [0, n) on up to `nthreads` threads (slot < nthreads identifies the thread)
    void run(int n, int nthreads, const std::function<void(int, int)> &fn) {
        if (n <= 0) return;
        if (nthreads > n) nthreads = n;
        if (nthreads <= 1) { for (int i = 0; i < n; ++i) fn(i, 0); return; }
        std::unique_lock<std::mutex> call_lock(call_mu_);   // one job at a time
        ensure(nthreads - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); active_ = nthreads - 1; pending_ = nthreads - 1; ++gen_;
        }
        cv_.notify_all();
        work(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
  private:
    Pool() {}
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void ensure(int nworkers) {
        while ((int)threads_.size() < nworkers) {
            const int id = (int)threads_.size();
            threads_.emplace_back([this, id] { loop(id); });
        }
    }
    void work(int slot) {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i, slot);
        }
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            if (id >= active_) continue;     // not needed for this job
            lk.unlock();
            work(id + 1);
            lk.lock();
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> threads_;
    const std::function<void(int, int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, active_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

struct Philox {
    uint32_t key[2];
    uint32_t ctr[4];
    uint32_t out[4];
    int have = 0;
    double spare = 0.0;
    bool has_spare = false;
    Philox(uint64_t seed, uint32_t id, uint32_t epoch, uint32_t stream) {
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
        ctr[0] = 0; ctr[1] = id; ctr[2] = epoch; ctr[3] = stream;
    }
    void refill() {
        uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
        ctr[0] += 1;  // 2^32 blocks per (id, epoch, stream): ample
        have = 4;
    }
    uint32_t u32() { if (!have) refill(); return out[--have]; }
    double uniform() {  // (0,1), 53 bits
        const uint64_t a = u32(), b = u32();
        return ((double)(((a << 32) | b) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    }
    double normal() {
        if (has_spare) { has_spare = false; return spare; }
        const double u1 = uniform(), u2 = uniform();
        const double r = sqrt(-2.0 * log(u1));
        double s, c;
        sincos(6.283185307179586476925 * u2, &s, &c);
        spare = r * s; has_spare = true;
        return r * c;
    }
    double gamma(double a) {  // Marsaglia-Tsang, shape a > 0, scale 1
        if (a < 1.0) {
            const double u = uniform();
            return gamma(a + 1.0) * pow(u, 1.0 / a);
        }
        const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
        for (;;) {
            double x, v;
            do { x = normal(); v = 1.0 + c * x; } while (v <= 0.0);
            v = v * v * v;
            const double u = uniform();
            if (u < 1.0 - 0.0331 * x * x * x * x) return d * v;
            if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return d * v;
        }
    }
};

// Psi (row-major, symmetric, D x D) = U U', U upper triangular (row-major, zeros below the diagonal).
// Returns false when Psi is not positive definite.
bool reverse_cholesky(const double *P, int D, double *U) {
    memset(U, 0, sizeof(double) * (size_t)D * D);
    for (int j = D - 1; j >= 0; --j) {
        double s = P[(size_t)j * D + j];
        const double *uj = U + (size_t)j * D;
#pragma omp simd reduction(- : s)
        for (int k = j + 1; k < D; ++k) s -= uj[k] * uj[k];
        if (!(s > 0.0)) return false;
        const double ujj = sqrt(s);
        U[(size_t)j * D + j] = ujj;
        const double inv = 1.0 / ujj;
        for (int i = 0; i < j; ++i) {
            double t = P[(size_t)i * D + j];
            const double *ui = U + (size_t)i * D;
#pragma omp simd reduction(- : t)
            for (int k = j + 1; k < D; ++k) t -= ui[k] * uj[k];
            U[(size_t)i * D + j] = t * inv;
        }
    }
    return true;
}

// priors/niw.jl:20-31 for one statistic set; psi_out symmetric.  N == 0 -> prior.
void niw_posterior_one(int D, double k0, double v0, const double *m0, const double *psi0, double N, const double *sum,
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

}  // namespace

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

// Fused unpack + posterior for the clusters in `sel` (NULL: all K), in place on the sampler's persistent
// arrays.  packed rows follow include/dpmm_hip.h: row 2k+s = {N, sum[D], lower triangle of S}.
// Outputs, per distribution row 3k+w (w = 0 cluster = left + right, 1 left, 2 right):
//   N[3K], sums[3K][D], S[3K][D*D], kappa/nu[3K], m[3K][D], U[3K][D*D], logdet_psi[3K].
// Replaces update_suff_stats_posterior!'s per-cluster aggregate + update_splittable_cluster_params!
// (src/local_clusters_actions.jl:237-251,137-147).
HAPI int dpmmh_niw_update_from_packed(int K, int D, const double *packed, int64_t stride, const int32_t *sel, int nsel,
                                      double kappa0, double nu0, const double *m0, const double *psi0, double *N,
                                      double *sums, double *S, double *kappa, double *nu, double *m, double *U,
                                      double *logdet_psi, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    const int nk = sel ? nsel : K;
    const size_t DD = (size_t)D * D;
    {
        std::vector<std::vector<double>> scratch(nthreads, std::vector<double>(2 * DD));
        Pool::get().run(3 * nk, nthreads, [&](int item, int slot) {
            {
                const int j = item / 3, w = item % 3;
                double *P = scratch[slot].data(), *psi_ = P + DD;
                struct { double *p; double *data() { return p; } double &operator[](size_t e) { return p[e]; } } psi{psi_};
                const int k = sel ? sel[j] : j;
                const double *l = packed + (size_t)(2 * k) * stride, *r = l + stride;
                const int row = 3 * k + w;
                double *Sr = S + (size_t)row * DD, *sr = sums + (size_t)row * D;
                const double cl = (w != 2) ? 1.0 : 0.0, cr = (w != 1) ? 1.0 : 0.0;   // c = l + r
                N[row] = cl * l[0] + cr * r[0];
                for (int d = 0; d < D; ++d) sr[d] = cl * l[1 + d] + cr * r[1 + d];
                const double *tl = l + 1 + D, *tr = r + 1 + D;
                for (int a = 0; a < D; ++a)
                    for (int b = 0; b <= a; ++b) {
                        const size_t t = (size_t)a * (a + 1) / 2 + b;
                        const double v = cl * tl[t] + cr * tr[t];
                        Sr[(size_t)a * D + b] = v;
                        Sr[(size_t)b * D + a] = v;
                    }
                niw_posterior_one(D, kappa0, nu0, m0, psi0, N[row], sr, Sr, &kappa[row], &nu[row], m + (size_t)row * D, psi.data());
                for (size_t e = 0; e < DD; ++e) P[e] = psi[e] * nu[row];
                double *Uo = U + (size_t)row * DD;
                if (reverse_cholesky(P, D, Uo)) {
                    double ld = 0.0;
                    for (int d = 0; d < D; ++d) ld += log(Uo[(size_t)d * D + d]);
                    logdet_psi[row] = 2.0 * ld - D * log(nu[row]);
                } else {
                    logdet_psi[row] = NAN;
                }
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
        Philox rng(seed, (uint32_t)ids[i], epoch, 16u);
        double *A = A_noise + (size_t)i * D * D;
        for (int r = 0; r < D; ++r)
            for (int c = 0; c < r; ++c) A[(size_t)r * D + c] = rng.normal();
        for (int d = 0; d < D; ++d) xi[(size_t)i * D + d] = rng.normal();
    });
    return 0;
}

static int niw_sample_impl(int n, int D, const double *kappa, const double *nu, const double *m, const double *U,
                           uint64_t seed, uint32_t epoch, const int32_t *ids, const double *A_noise, const double *xi_in,
                           float *mu, float *R, float *logdet_sigma, int nthreads);

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

static int niw_sample_impl(int n, int D, const double *kappa, const double *nu, const double *m, const double *U,
                           uint64_t seed, uint32_t epoch, const int32_t *ids, const double *A_noise, const double *xi_in,
                           float *mu, float *R, float *logdet_sigma, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    {
        const size_t DD = (size_t)D * D;
        std::vector<std::vector<double>> scratch(nthreads, std::vector<double>(2 * DD + 3 * (size_t)D));
        std::vector<std::vector<double>> blk(nthreads, std::vector<double>(8 * (size_t)D));
        Pool::get().run(n, nthreads, [&](int i, int slot) {
            struct V { double *p; double *data() { return p; } double &operator[](size_t e) { return p[e]; } };
            double *base = scratch[slot].data();
            V A{base}, Rl{base + DD}, a{base + 2 * DD}, xi{base + 2 * DD + D}, v{base + 2 * DD + 2 * D};
            (void)a;
            // normals: stream 16 (identical whether pre-generated or not); chi-squares: stream 18
            Philox rng(seed, (uint32_t)ids[i], epoch, 16u), rng_chi(seed, (uint32_t)ids[i], epoch, 18u);
            const double *Ui = U + (size_t)i * D * D;
            // Bartlett factor, lower triangular (A[r][c], r >= c)
            const double *An = A_noise ? A_noise + (size_t)i * D * D : nullptr;
            for (int r = 0; r < D; ++r) {
                for (int c = 0; c < r; ++c) A[(size_t)r * D + c] = An ? An[(size_t)r * D + c] : rng.normal();
                A[(size_t)r * D + r] = sqrt(2.0 * rng_chi.gamma(0.5 * (nu[i] - r)));
            }
            // R = A' U^-1 : row j of R solves r_j U[j:, j:] = A[j:, j]'
            // JB rows of R at a time share every pass over a row of U (at D = 256 U is 512 KiB: one row of R per pass was
            // bound by streaming U from L2).  Per row the operations and their order are those of the one-row loop: same bits.
            memset(Rl.data(), 0, sizeof(double) * DD);
            double ld = 0.0;
            constexpr int JB = 8;
            for (int j0 = 0; j0 < D; j0 += JB) {
                const int nb = std::min(JB, D - j0);
                double *ab = blk[slot].data();                        // [JB][D], row jb = column j0 + jb of A (zero above the diagonal)
                for (int jb = 0; jb < nb; ++jb) {
                    double *ar = ab + (size_t)jb * D;
                    for (int r = 0; r < j0 + jb; ++r) ar[r] = 0.0;
                    for (int r = j0 + jb; r < D; ++r) ar[r] = A[(size_t)r * D + j0 + jb];
                }
                for (int c = j0; c < D; ++c) {
                    const double *uc = Ui + (size_t)c * D;
                    const double ucc = uc[c];
                    double val[JB];
                    for (int jb = 0; jb < nb; ++jb) {
                        // rows that have not started yet (c < j0 + jb) hold 0 here: val = 0 and the update below is a no-op
                        val[jb] = (c >= j0 + jb) ? ab[(size_t)jb * D + c] / ucc : 0.0;
                        if (c >= j0 + jb) Rl[(size_t)(j0 + jb) * D + c] = val[jb];
                    }
                    for (int jb = 0; jb < nb; ++jb) {
                        if (c < j0 + jb) continue;
                        double *ap = ab + (size_t)jb * D;
                        const double v = val[jb];
#pragma omp simd
                        for (int cc = c + 1; cc < D; ++cc) ap[cc] -= v * uc[cc];
                    }
                }
                for (int jb = 0; jb < nb; ++jb) ld += log(Rl[(size_t)(j0 + jb) * D + j0 + jb]);
            }
            logdet_sigma[i] = (float)(-2.0 * ld);
            // mu = m + R^-1 xi / sqrt(kappa)
            for (int d = 0; d < D; ++d) xi[d] = xi_in ? xi_in[(size_t)i * D + d] : rng.normal();
            for (int r = D - 1; r >= 0; --r) {
                double s = xi[r];
                const double *rr = Rl.data() + (size_t)r * D;
                const double *vp = v.data();
#pragma omp simd reduction(- : s)
                for (int c = r + 1; c < D; ++c) s -= rr[c] * vp[c];
                v[r] = s / rr[r];
            }
            const double isk = 1.0 / sqrt(kappa[i]);
            for (int d = 0; d < D; ++d) mu[(size_t)i * D + d] = (float)(m[(size_t)i * D + d] + v[d] * isk);
            float *Ro = R + (size_t)i * D * D;
            for (size_t e = 0; e < (size_t)D * D; ++e) Ro[e] = (float)Rl[e];
        });
    }
    return 0;
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
    {
        std::vector<std::vector<double>> scratch(nthreads, std::vector<double>(D));
        Pool::get().run(n, nthreads, [&](int i, int slot) {
            double *lg = scratch[slot].data();
            Philox rng(seed, (uint32_t)ids[i], epoch, 17u);
            const float *al = alpha + (size_t)i * D;
            // work with log-gammas so that tiny shapes do not underflow: log g = log Gamma(a+1) draw + log(u)/a
            double mx = -INFINITY;
            for (int d = 0; d < D; ++d) {
                const double a = (double)al[d];
                double l;
                if (a < 1.0) l = log(rng.gamma(a + 1.0)) + log(rng.uniform()) / a;
                else l = log(rng.gamma(a));
                lg[d] = l;
                if (l > mx) mx = l;
            }
            double s = 0.0;
            for (int d = 0; d < D; ++d) s += exp(lg[d] - mx);
            const double lse = mx + log(s);
            for (int d = 0; d < D; ++d) logp[(size_t)i * D + d] = (float)(lg[d] - lse);
        });
    }
    return 0;
}

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
