// mult_master.hip -- the Multinomial master's parameter draws on the device.
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

hipError_t launch_mult_dirichlet(const double *rows, int64_t stride, const float *alpha0, const float *alpha1, int outlier_first, int D, int64_t ldx,
                                 int K, uint64_t seed, uint32_t epoch, float *raw, hipStream_t s) {
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void *)mult_dirichlet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DPMM_MULT_MASTER_MAXD); attr = true; }
    hipLaunchKernelGGL(mult_dirichlet_kernel, dim3(3 * K), dim3(256), sizeof(double) * (size_t)D, s, rows, stride, alpha0, alpha1, outlier_first, D, ldx,
                       seed, epoch, raw);
    return hipGetLastError();
}

}  // namespace dpmm
