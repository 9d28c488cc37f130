// mult_master.hip -- the Multinomial master's parameter draws and log-marginals on the device.
//
// Stands in for sample_distribution of the Multinomial prior (src/priors/multinomial_prior.jl:23-25: log.(rand(Dirichlet(alpha')))) with
// calc_posterior in front of it (:16-21: alpha' = alpha + sum x, the sum held as Float32) for all 3K distributions of a sweep: at D = 1000,
// K = 32 the host needs ~0.08 ms of 14 threads for the 96 000 Gamma variates plus a 384 KB gather over the host link for the
// hand-over; here the packed statistics rows are read where the statistics pass left them and the log-probabilities are born in the
// layout the pack kernels read (raw [3K][ldx] Float32).
//   row j = 3 k + w of cluster k: w = 0 cluster (left + right), 1 left, 2 right; rows of the pass: left = rows[2k], right = rows[2k+1],
//   each {N, sum x_1 .. sum x_D} Float64.  alpha'_d = alpha_d + Float32(sum) (N = 0: the prior itself, as calc_posterior does).
//   g_d ~ Gamma(alpha'_d, 1) by Marsaglia-Tsang in Float64 (a < 1: Gamma(a + 1) U^(1/a), carried in logs so that tiny components do not
//   underflow), log p_d = log g_d - logsumexp(log g).  Streams: Philox(seed; (row << 32) + 64 d + trial, epoch, STREAM_MULT_DIR) -- keyed by
//   the position in cluster order like the NIW draws, so every rank of a multi-GPU run draws the same parameters from the same rows.
#include "dpmm_device.h"
#include "dpmm_kernels.h"

namespace dpmm {

enum : uint32_t { STREAM_MULT_DIR = 35 };

__device__ __forceinline__ double mm_u53(uint32_t a, uint32_t b) { return ((double)((((uint64_t)a << 32) | b) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

// log of a Gamma(a, 1) variate, a > 0
__device__ double log_gamma_variate(double a, uint64_t seed, uint64_t base, uint32_t epoch) {
    double boost = 0.0;
    if (a < 1.0) {
        const Philox4 ru = philox4x32_10(seed, base + 63, epoch, STREAM_MULT_DIR);
        boost = log(mm_u53(ru.v[0], ru.v[1])) / a;
        a += 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0;; ++t) {
        const Philox4 rn = philox4x32_10(seed, base + 2 * t, epoch, STREAM_MULT_DIR);
        const double u1 = mm_u53(rn.v[0], rn.v[1]), u2 = mm_u53(rn.v[2], rn.v[3]);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
        const Philox4 ru = philox4x32_10(seed, base + 2 * t + 1, epoch, STREAM_MULT_DIR);
        const double u = mm_u53(ru.v[0], ru.v[1]);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (u < 1.0 - 0.0331 * x * x * x * x || log(u) < 0.5 * x * x + d * (1.0 - v + log(v)) || t > 28) return log(d * v) + boost;
    }
}

__global__ __launch_bounds__(256) void mult_dirichlet_kernel(const double *__restrict__ rows, int64_t stride, const float *__restrict__ alpha0,
                                                             const float *__restrict__ alpha1, int outlier_first, int D, int64_t ldx,
                                                             uint64_t seed, uint32_t epoch, float *__restrict__ raw) {
    extern __shared__ double lg[];             // [D] log Gamma variates of this row
    __shared__ double red[256];
    const int j = blockIdx.x, k = j / 3, w = j % 3, tid = threadIdx.x;
    const double *l = rows + (int64_t)(2 * k) * stride, *r = l + stride;
    const double cl = (w != 2) ? 1.0 : 0.0, cr = (w != 1) ? 1.0 : 0.0;
    const double N = cl * l[0] + cr * r[0];
    const float *alpha = (outlier_first && k == 0 && alpha1) ? alpha1 : alpha0;
    double mx = -INFINITY;
    for (int d = tid; d < D; d += 256) {
        const float a = (N == 0.0) ? alpha[d] : alpha[d] + (float)(cl * l[1 + d] + cr * r[1 + d]);
        const double v = log_gamma_variate((double)a, seed, ((uint64_t)j << 32) + 64ull * (uint64_t)d, epoch);
        lg[d] = v;
        mx = fmax(mx, v);
    }
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmax(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    double s = 0.0;
    for (int d = tid; d < D; d += 256) s += exp(lg[d] - mx);
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double lse = mx + log(red[0]);
    float *out = raw + (int64_t)j * ldx;
    for (int d = tid; d < (int)ldx; d += 256) out[d] = d < D ? (float)(lg[d] - lse) : 0.f;
}

// Log-marginals of the Multinomial master (src/priors/multinomial_prior.jl:34-39: lgamma(sum a) - lgamma(sum a') + sum (lgamma(a'_d) - lgamma(a_d)),
// a' = a + Float32(sum x) as calc_posterior forms it) for the 3K distributions of a pass and for the POOLED statistics of cluster pairs
// (check_and_merge!, src/local_clusters_actions.jl:385-413: left + right of both clusters).  One workgroup per item, Float64 lgamma, tree sums.
//   item < 3K: distribution 3k + w -> out[2 item] = N, out[2 item + 1] = log-marginal (0 for N = 0: posterior = prior);
//   item = 3K + p: pair (pairs[2p], pairs[2p+1]) -> out[6K + p].   prior_c = {sum a, sum lgamma(a)} of the cluster prior | the outlier prior
__global__ __launch_bounds__(256) void mult_marginal_kernel(const double *__restrict__ rows, int64_t stride, const float *__restrict__ alpha0,
                                                            const float *__restrict__ alpha1, int outlier_first, int D, int K,
                                                            const int32_t *__restrict__ pairs, double a0_sum, double a0_lg, double a1_sum, double a1_lg,
                                                            double *__restrict__ out) {
    __shared__ double r1[256], r2[256];
    const int item = blockIdx.x, tid = threadIdx.x;
    const double *src[4] = {nullptr, nullptr, nullptr, nullptr};
    double coef[4] = {0.0, 0.0, 0.0, 0.0};
    int kprior;
    if (item < 3 * K) {
        const int k = item / 3, w = item % 3;
        src[0] = rows + (int64_t)(2 * k) * stride; src[1] = src[0] + stride;
        coef[0] = (w != 2) ? 1.0 : 0.0; coef[1] = (w != 1) ? 1.0 : 0.0;
        src[2] = src[0]; src[3] = src[0];
        kprior = k;
    } else {
        const int p = item - 3 * K, i = pairs[2 * p], j = pairs[2 * p + 1];
        src[0] = rows + (int64_t)(2 * i) * stride; src[1] = src[0] + stride;
        src[2] = rows + (int64_t)(2 * j) * stride; src[3] = src[2] + stride;
        coef[0] = coef[1] = coef[2] = coef[3] = 1.0;
        kprior = i;
    }
    const bool outl = outlier_first && kprior == 0 && alpha1;
    const float *alpha = outl ? alpha1 : alpha0;
    const double N = coef[0] * src[0][0] + coef[1] * src[1][0] + coef[2] * src[2][0] + coef[3] * src[3][0];
    double s1 = 0.0, acc = 0.0;
    if (N != 0.0)
        for (int d = tid; d < D; d += 256) {
            const float a = alpha[d] + (float)(coef[0] * src[0][1 + d] + coef[1] * src[1][1 + d] + coef[2] * src[2][1 + d] + coef[3] * src[3][1 + d]);
            s1 += (double)a;
            acc += lgamma((double)a);
        }
    r1[tid] = s1; r2[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { r1[tid] += r1[tid + o]; r2[tid] += r2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double L = (N == 0.0) ? 0.0 : lgamma(outl ? a1_sum : a0_sum) - lgamma(r1[0]) + (r2[0] - (outl ? a1_lg : a0_lg));
        if (item < 3 * K) { out[2 * item] = N; out[2 * item + 1] = L; }
        else out[6 * K + (item - 3 * K)] = L;
    }
}

hipError_t launch_mult_marginals(const double *rows, int64_t stride, const float *alpha0, const float *alpha1, int outlier_first, int D, int K,
                                 const int32_t *pairs, int npairs, const double prior_c[4], double *out, hipStream_t s) {
    DPMM_LAUNCH(mult_marginal_kernel, dim3(3 * K + npairs), dim3(256), 0, s, rows, stride, alpha0, alpha1, outlier_first, D, K, pairs,
                       prior_c[0], prior_c[1], prior_c[2], prior_c[3], out);
    return hipGetLastError();
}

hipError_t launch_mult_dirichlet(const double *rows, int64_t stride, const float *alpha0, const float *alpha1, int outlier_first, int D, int64_t ldx,
                                 int K, uint64_t seed, uint32_t epoch, float *raw, hipStream_t s) {
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void *)mult_dirichlet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DPMM_MULT_MASTER_MAXD); attr = true; }
    DPMM_LAUNCH(mult_dirichlet_kernel, dim3(3 * K), dim3(256), sizeof(double) * (size_t)D, s, rows, stride, alpha0, alpha1, outlier_first, D, ldx,
                       seed, epoch, raw);
    return hipGetLastError();
}

}  // namespace dpmm
