// Dev microbenchmark: do the FP64 matrix instruction (v_mfma_f64_16x16x4_f64) and FP64 vector FMAs (v_fma_f64) run on separate
// pipes of a SIMD (rates add) or on the same units (rates do not add)?  Decides whether the NIW statistics kernel could split its
// outer-product work between the two.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// mode 0: MFMA only; 1: vector FMA only; 2: both interleaved in every wave; 3: even waves MFMA, odd waves FMA
template <int MODE>
__global__ __launch_bounds__(256) void kern(double *out, int iters, double a0) {
    f64x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f64x4){0, 0, 0, 0};
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = a0 * i;
    const double a = a0 + threadIdx.x * 1e-3, b = a0 * 0.5 + threadIdx.x * 2e-3;
    const bool mf = MODE == 0 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 0);
    const bool vf = MODE == 1 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (mf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
            if (vf) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], a, b);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
float run(F launch, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(iters); hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    double *out; hipMalloc(&out, 4096 * 256 * 8);
    const int iters = 2000;
    for (int wg : {256, 512}) {
        // per block-iteration: MFMA 16 x (16*16*4*2 = 2048 flops) per wave x 4 waves ; FMA 4 x 16 x 64 lanes x 2 flops per wave x 4 waves
        const double f_m = 4.0 * 16 * 2048, f_v = 4.0 * 4 * 16 * 64 * 2;
        float t0 = run([&](int it) { kern<0><<<wg, 256>>>(out, it, 1.0); }, iters);
        float t1 = run([&](int it) { kern<1><<<wg, 256>>>(out, it, 1.0); }, iters);
        float t2 = run([&](int it) { kern<2><<<wg, 256>>>(out, it, 1.0); }, iters);
        float t3 = run([&](int it) { kern<3><<<wg, 256>>>(out, it, 1.0); }, iters);
        printf("blocks=%d: MFMA only %.3f ms = %.1f TF | FMA only %.3f ms = %.1f TF | both per wave %.3f ms = %.1f TF (sum of the two alone %.3f ms) | waves split %.3f ms = %.1f TF\n",
               wg, t0, f_m * wg * iters / (t0 * 1e-3) / 1e12, t1, f_v * wg * iters / (t1 * 1e-3) / 1e12, t2, (f_m + f_v) * wg * iters / (t2 * 1e-3) / 1e12, t0 + t1,
               t3, (f_m + f_v) * 0.5 * wg * iters / (t3 * 1e-3) / 1e12);
    }
    return 0;
}
