// Dev microbenchmark: sustained rate of the f32-input MFMAs with register operands only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char *name, F launch, double flops_per_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(iters);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-28s %.3f ms  %.1f TFLOP/s\n", name, best, flops_per_block_iter * blocks * iters / (best * 1e-3) / 1e12);
}
int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 4000;
    for (int wg : {256, 512, 1024}) {
        printf("blocks=%d (x256 threads)\n", wg);
        run("16x16x4 1acc", [&](int it) { k16<1><<<wg, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 1 * 2048, wg, iters);
        run("16x16x4 2acc", [&](int it) { k16<2><<<wg, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 2 * 2048, wg, iters);
        run("16x16x4 4acc", [&](int it) { k16<4><<<wg, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 4 * 2048, wg, iters);
        run("32x32x2 1acc", [&](int it) { k32<1><<<wg, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 1 * 4096, wg, iters);
        run("32x32x2 2acc", [&](int it) { k32<2><<<wg, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 2 * 4096, wg, iters);
        run("32x32x2 4acc", [&](int it) { k32<4><<<wg, 256>>>(out, it, 1.f, 2.f); }, 4.0 * 8 * 4 * 4096, wg, iters);
    }
    return 0;
}
