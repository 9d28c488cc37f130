/*
 * dpmm_oracle.c -- CPU restatement of the DPMMSubClusters.jl worker path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libdpmmhip.so + the host
 * package) never links, imports or calls anything in oracle/.
 *
 * Parity status: the reference is pure Julia and cannot be executed in the build
 * container (no julia), and it ships no value-level tests for the sampler.  What IS
 * pinned against reference artefacts (tests/test_oracle_golden.py):
 *   - Multinomial sufficient statistics + posterior: bit-exact against the vectors
 *     stored in test/save_load_test/checkpoint_20.jld2 (tests/golden/mnm_golden.npz)
 *   - NIW sufficient statistics + posterior: <=1e-9 rel against
 *     examples/save_load_model/checkpoint__50.jld2 (tests/golden/niw_golden.npz)
 * The random-number stream (Julia's global RNG through un-pinned StatsBase/Distributions)
 * is "parity unpinned": the build defines its own counter-based stream (Philox4x32-10)
 * shared bit-for-bit by this oracle and the HIP kernels.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout, src/...).  Written from the behavioural description, in C,
 * not translated line by line.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -fopenmp  (see oracle/Makefile)
 * -ffp-contract=off matters: the draw arithmetic must round exactly like the HIP
 * kernel's explicit fmaf/add sequence.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Counter-based RNG shared with the HIP kernels (build-defined; the reference
 * seeds Julia's global RNG identically on every process, dp-parallel-sampling.jl:37-39). */

static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k[0];
    const uint32_t n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

/* key = seed (lo,hi); counter = (idx lo, idx hi, epoch, stream) */
ORC_API void orc_philox(uint64_t seed, uint64_t idx, uint32_t epoch, uint32_t stream, uint32_t out[4]) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), epoch, stream};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

/* streams */
enum { ORC_STREAM_SWEEP = 0, ORC_STREAM_INIT = 1, ORC_STREAM_SPLIT = 2, ORC_STREAM_RESET = 3 };

/* (0, 1): odd multiples of 2^-24 (never 0: a threshold t = u * sum of exactly 0 would stop the scan at index 0 whatever its probability) */
static inline float u01(uint32_t r) { return (float)((r >> 8) | 1u) * (1.0f / 16777216.0f); }

ORC_API void orc_uniforms(uint64_t seed, uint32_t epoch, uint32_t stream, int64_t first_idx, int64_t n,
                          float *u0, float *u1) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t o[4];
        orc_philox(seed, (uint64_t)(first_idx + i), epoch, stream, o);
        if (u0) u0[i] = u01(o[0]);
        if (u1) u1[i] = u01(o[1]);
    }
}

/* ------------------------------------------------------------------------- */
/* Deterministic expf: identical operation sequence in the HIP kernel
 * (csrc/dpmm_device.h: exp_det).  Used only for max-shifted arguments (x <= 0). */
static inline float exp_det(float x) {
    if (!(x >= -86.0f)) return 0.0f; /* also maps NaN to 0 */
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return p * s.f;
}

ORC_API float orc_exp_det(float x) { return exp_det(x); }

/* ------------------------------------------------------------------------- */
/* Log-likelihoods.  X is D x n column-major with leading dimension ldx
 * (point i = X + i*ldx), as `points` in ds.jl:53. */

/* distributions/mv_gaussian.jl:21-25 (+ utils.jl:75-84): z = x - mu; r = z . (invS z);
 * out = -((length(Sigma)*log(2pi) + logdet)/2) - r/2, all Float32.  NB length(Sigma) = D*D. */
ORC_API void orc_niw_loglik_ref(const float *X, int D, int64_t n, int64_t ldx, const float *mu,
                                const float *invS /* D x D column-major */, float logdet, float *out) {
    const float cst = -(((float)(D * D) * (float)log(2.0 * M_PI) + logdet) / 2.0f);
#pragma omp parallel
    {
        float *z = (float *)malloc(sizeof(float) * (size_t)D);
        float *y = (float *)malloc(sizeof(float) * (size_t)D);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            const float *x = X + i * ldx;
            for (int d = 0; d < D; ++d) { z[d] = x[d] - mu[d]; y[d] = 0.0f; }
            for (int k = 0; k < D; ++k) { /* y = invS * z, column sweep (gemm order is unspecified in the reference) */
                const float zk = z[k];
                const float *col = invS + (size_t)k * D;
                for (int d = 0; d < D; ++d) y[d] += col[d] * zk;
            }
            float r = 0.0f;
            for (int d = 0; d < D; ++d) r += z[d] * y[d];
            out[i] = cst - r / 2.0f;
        }
        free(z); free(y);
    }
}

/* Same quantity in double precision without the D*D quirk shift folded differently:
 * returns the exact quadratic form and constant separately for margin analysis. */
ORC_API void orc_niw_loglik_f64(const float *X, int D, int64_t n, int64_t ldx, const float *mu,
                                const float *invS, float logdet, double *out) {
    const double cst = -(((double)D * D * log(2.0 * M_PI) + (double)logdet) / 2.0);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *x = X + i * ldx;
        double r = 0.0;
        for (int a = 0; a < D; ++a) {
            const double za = (double)x[a] - (double)mu[a];
            double ya = 0.0;
            for (int b = 0; b < D; ++b) ya += (double)invS[(size_t)b * D + a] * ((double)x[b] - (double)mu[b]);
            r += za * ya;
        }
        out[i] = cst - r / 2.0;
    }
}

/* distributions/multinomial_dist.jl:13-15: r_i = sum_d alpha_d * x_{d,i} (alpha = log-probs) */
ORC_API void orc_mult_loglik_ref(const float *X, int D, int64_t n, int64_t ldx, const float *logp, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *x = X + i * ldx;
        float r = 0.0f;
        for (int d = 0; d < D; ++d) r += logp[d] * x[d];
        out[i] = r;
    }
}

/* ------------------------------------------------------------------------- */
/* utils.jl:19-31 sample_log_cat_array! : parr is [K][n] (= Julia's column-major n x K).
 * NaN -> -Inf; subtract row max; exp; (division by the row sum is folded into the
 * threshold t = u * sum, same draw); inverse-CDF linear scan `cw < t && i < n`
 * (StatsBase.sample with ProbabilityWeights, one uniform per row).
 * Row of all -Inf -> NaN weights in the reference -> scan never advances -> index 1.
 * Labels are 1-based Int64 (ds.jl:54-55). */
static inline int64_t draw_row(const float *parr, int64_t n, int64_t i, int K, float u) {
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) {
        float a = parr[(size_t)k * n + i];
        if (a != a) a = -INFINITY;
        if (a > m) m = a;
    }
    if (m == -INFINITY) return 1;
    float s = 0.0f;
    for (int k = 0; k < K; ++k) {
        float a = parr[(size_t)k * n + i];
        if (a != a) a = -INFINITY;
        s += exp_det(a - m);
    }
    const float t = u * s;
    float cw = 0.0f;
    int64_t lab = K;
    for (int k = 0; k < K; ++k) {
        float a = parr[(size_t)k * n + i];
        if (a != a) a = -INFINITY;
        cw += exp_det(a - m);
        if (!(cw < t)) { lab = k + 1; break; }
    }
    return lab;
}

ORC_API void orc_sample_log_cat(const float *parr, int64_t n, int K, const float *u, int64_t *labels) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) labels[i] = draw_row(parr, n, i, K, u[i]);
}

/* local_clusters_actions.jl:129-130: argmax per row, first maximum wins; Julia's argmax
 * returns the first NaN if any is present (no NaN guard on this path). */
ORC_API void orc_argmax_rows(const float *parr, int64_t n, int K, int64_t *labels) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int best = 0;
        float m = parr[i];
        int nan_seen = (m != m);
        for (int k = 1; k < K && !nan_seen; ++k) {
            const float a = parr[(size_t)k * n + i];
            if (a != a) { best = k; nan_seen = 1; }
            else if (a > m) { m = a; best = k; }
        }
        labels[i] = best + 1;
    }
}

/* ------------------------------------------------------------------------- */
/* One worker's label + sub-label sampling for a shard.
 * local_clusters_actions.jl:112-134 (sample_labels_worker!) then :70-95
 * (sample_sub_clusters_worker!/create_subclusters_labels!).
 * Params: mu [3K][D], invS [3K][D*D] col-major, logdet [3K] ordered (cluster, left, right)
 * per cluster k: rows 3k, 3k+1, 3k+2; logw [K] = log(weights) ; loglr [K][2] = log(lr_weights).
 * first_idx = global index of the shard's first point (RNG counter).
 * parr_out (optional) receives the [K][n] table of loglik + log w (Float32). */
ORC_API void orc_sweep_niw(const float *X, int D, int64_t n, int64_t ldx, int K, const float *mu,
                           const float *invS, const float *logdet, const float *logw, const float *loglr,
                           uint64_t seed, uint32_t epoch, int64_t first_idx, int final_argmax,
                           int64_t *labels, int64_t *sub, float *parr_out) {
    float *parr = parr_out ? parr_out : (float *)malloc(sizeof(float) * (size_t)K * (size_t)n);
    for (int k = 0; k < K; ++k) {
        float *col = parr + (size_t)k * n;
        orc_niw_loglik_ref(X, D, n, ldx, mu + (size_t)(3 * k) * D, invS + (size_t)(3 * k) * D * D, logdet[3 * k], col);
        for (int64_t i = 0; i < n; ++i) col[i] += logw[k];
    }
    float *u0 = (float *)malloc(sizeof(float) * (size_t)n), *u1 = (float *)malloc(sizeof(float) * (size_t)n);
    orc_uniforms(seed, epoch, ORC_STREAM_SWEEP, first_idx, n, u0, u1);
    if (final_argmax) orc_argmax_rows(parr, n, K, labels);
    else orc_sample_log_cat(parr, n, K, u0, labels);
    /* sub-labels: always sampled (never argmax), :83-95 */
    float *p2 = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < K; ++k) {
        /* evaluate l and r for every point, use only the rows with label k+1 */
        for (int s = 0; s < 2; ++s) {
            orc_niw_loglik_ref(X, D, n, ldx, mu + (size_t)(3 * k + 1 + s) * D, invS + (size_t)(3 * k + 1 + s) * D * D,
                               logdet[3 * k + 1 + s], tmp);
            for (int64_t i = 0; i < n; ++i)
                if (labels[i] == k + 1) p2[(size_t)s * n + i] = tmp[i] + loglr[2 * k + s];
        }
    }
    orc_sample_log_cat(p2, n, 2, u1, sub);
    free(p2); free(tmp); free(u0); free(u1);
    if (!parr_out) free(parr);
}

ORC_API void orc_sweep_mult(const float *X, int D, int64_t n, int64_t ldx, int K, const float *logp /* [3K][D] */,
                            const float *logw, const float *loglr, uint64_t seed, uint32_t epoch,
                            int64_t first_idx, int final_argmax, int64_t *labels, int64_t *sub, float *parr_out) {
    float *parr = parr_out ? parr_out : (float *)malloc(sizeof(float) * (size_t)K * (size_t)n);
    for (int k = 0; k < K; ++k) {
        float *col = parr + (size_t)k * n;
        orc_mult_loglik_ref(X, D, n, ldx, logp + (size_t)(3 * k) * D, col);
        for (int64_t i = 0; i < n; ++i) col[i] += logw[k];
    }
    float *u0 = (float *)malloc(sizeof(float) * (size_t)n), *u1 = (float *)malloc(sizeof(float) * (size_t)n);
    orc_uniforms(seed, epoch, ORC_STREAM_SWEEP, first_idx, n, u0, u1);
    if (final_argmax) orc_argmax_rows(parr, n, K, labels);
    else orc_sample_log_cat(parr, n, K, u0, labels);
    float *p2 = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < K; ++k)
        for (int s = 0; s < 2; ++s) {
            orc_mult_loglik_ref(X, D, n, ldx, logp + (size_t)(3 * k + 1 + s) * D, tmp);
            for (int64_t i = 0; i < n; ++i)
                if (labels[i] == k + 1) p2[(size_t)s * n + i] = tmp[i] + loglr[2 * k + s];
        }
    orc_sample_log_cat(p2, n, 2, u1, sub);
    free(p2); free(tmp); free(u0); free(u1);
    if (!parr_out) free(parr);
}

/* ------------------------------------------------------------------------- */
/* Sufficient statistics.
 * local_clusters_actions.jl:149-169 (create_suff_stats_dict_worker) with
 * priors/niw.jl:42-51: pts -> Float64, points_sum = sum, S = pts*pts', S = 0.5(S+S').
 * Output order per cluster k (0-based row 3k + w): w=0 cluster, 1 left (sub==1), 2 right (sub==2).
 * The reference computes the cluster-level statistics from scratch (a third pass);
 * so does this restatement. */
ORC_API void orc_suffstats_niw(const float *X, int D, int64_t n, int64_t ldx, const int64_t *labels,
                               const int64_t *sub, int K, double *Nout /* [3K] */, double *sum /* [3K][D] */,
                               double *S /* [3K][D*D] col-major */) {
    memset(Nout, 0, sizeof(double) * 3 * (size_t)K);
    memset(sum, 0, sizeof(double) * 3 * (size_t)K * D);
    memset(S, 0, sizeof(double) * 3 * (size_t)K * D * D);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t k = labels[i] - 1;
        if (k < 0 || k >= K) continue;
        const float *x = X + i * ldx;
        const int rows[2] = {(int)(3 * k), (int)(3 * k + (sub[i] == 1 ? 1 : 2))};
        for (int w = 0; w < 2; ++w) {
            const int row = rows[w];
            Nout[row] += 1.0;
            double *sv = sum + (size_t)row * D;
            double *Sm = S + (size_t)row * D * D;
            for (int a = 0; a < D; ++a) {
                const double xa = (double)x[a];
                sv[a] += xa;
                for (int b = 0; b < D; ++b) Sm[(size_t)b * D + a] += xa * (double)x[b];
            }
        }
    }
    /* S = 0.5 (S + S') */
    for (int r = 0; r < 3 * K; ++r) {
        double *Sm = S + (size_t)r * D * D;
        for (int a = 0; a < D; ++a)
            for (int b = a + 1; b < D; ++b) {
                const double v = 0.5 * (Sm[(size_t)b * D + a] + Sm[(size_t)a * D + b]);
                Sm[(size_t)b * D + a] = v; Sm[(size_t)a * D + b] = v;
            }
    }
}

/* priors/multinomial_prior.jl:27-32: points_sum = sum(pts, dims=2) in Float32, N.
 * Julia's sum over dims=2 of a column-major D x n matrix adds the columns in order
 * (for each d: x[d,1] + x[d,2] + ...), which is what the golden fixture pins bit-exactly.
 * (The discarded pts*pts' at :30 is not reproduced.) */
ORC_API void orc_suffstats_mult(const float *X, int D, int64_t n, int64_t ldx, const int64_t *labels,
                                const int64_t *sub, int K, float *Nout /* [3K] */, float *sum /* [3K][D] */) {
    memset(Nout, 0, sizeof(float) * 3 * (size_t)K);
    memset(sum, 0, sizeof(float) * 3 * (size_t)K * D);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t k = labels[i] - 1;
        if (k < 0 || k >= K) continue;
        const float *x = X + i * ldx;
        const int rows[2] = {(int)(3 * k), (int)(3 * k + (sub[i] == 1 ? 1 : 2))};
        for (int w = 0; w < 2; ++w) {
            Nout[rows[w]] += 1.0f;
            float *sv = sum + (size_t)rows[w] * D;
            for (int d = 0; d < D; ++d) sv[d] += x[d];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Relabel operations (integer bookkeeping; parity bar = bit-exact).           */

static inline int64_t rand12(uint64_t seed, int64_t gidx, uint32_t epoch, uint32_t stream) {
    uint32_t o[4];
    orc_philox(seed, (uint64_t)gidx, epoch, stream, o);
    return 1 + (int64_t)(o[0] & 1u);
}

/* dp-parallel-sampling.jl:49-50: labels = rand(1:init_clusters), sub = rand(1:2) */
ORC_API void orc_init_labels(int64_t *labels, int64_t *sub, int64_t n, int init_clusters, uint64_t seed,
                             uint32_t epoch, int64_t first_idx) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t o[4];
        orc_philox(seed, (uint64_t)(first_idx + i), epoch, ORC_STREAM_INIT, o);
        labels[i] = 1 + (int64_t)(((uint64_t)o[0] * (uint64_t)init_clusters) >> 32);
        sub[i] = 1 + (int64_t)(o[1] & 1u);
    }
}

/* local_clusters_actions.jl:265-278 split_cluster_local_worker!: processed pair by pair, in order:
 * labels(label==idx & sub==2) = new_idx; then EVERY point that had label idx gets a fresh rand(1:2). */
ORC_API void orc_split_relabel(int64_t *labels, int64_t *sub, int64_t n, const int64_t *idx, const int64_t *new_idx,
                               int m, uint64_t seed, uint32_t epoch, int64_t first_idx) {
    for (int j = 0; j < m; ++j)
        for (int64_t i = 0; i < n; ++i)
            if (labels[i] == idx[j]) {
                if (sub[i] == 2) labels[i] = new_idx[j];
                sub[i] = rand12(seed, first_idx + i, epoch, ORC_STREAM_SPLIT);
            }
}

/* local_clusters_actions.jl:293-304 merge_clusters_worker!: pair by pair, in order */
ORC_API void orc_merge_relabel(int64_t *labels, int64_t *sub, int64_t n, const int64_t *idx, const int64_t *new_idx, int m) {
    for (int j = 0; j < m; ++j) {
        for (int64_t i = 0; i < n; ++i) if (labels[i] == idx[j]) sub[i] = 1;
        for (int64_t i = 0; i < n; ++i) if (labels[i] == new_idx[j]) { sub[i] = 2; labels[i] = idx[j]; }
    }
}

/* local_clusters_actions.jl:446-455 remove_empty_clusters_worker! */
ORC_API void orc_remove_empty(int64_t *labels, int64_t n, const int64_t *pts_count, int K) {
    int removed = 0;
    for (int k = 1; k <= K; ++k)
        if (pts_count[k - 1] == 0) {
            for (int64_t i = 0; i < n; ++i) if (labels[i] > k - removed) labels[i] -= 1;
            removed += 1;
        }
}

/* local_clusters_actions.jl:481-488 reset_bad_clusters_worker! (+ :474-479, :257-261 when idx==NULL => all) */
ORC_API void orc_reset_sub(const int64_t *labels, int64_t *sub, int64_t n, const int64_t *idx, int m, uint64_t seed,
                           uint32_t epoch, int64_t first_idx) {
    for (int64_t i = 0; i < n; ++i) {
        int hit = (idx == NULL);
        for (int j = 0; j < m && !hit; ++j) hit = (labels[i] == idx[j]);
        if (hit) sub[i] = rand12(seed, first_idx + i, epoch, ORC_STREAM_RESET);
    }
}
