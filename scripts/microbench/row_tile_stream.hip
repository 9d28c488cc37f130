// Microbenchmark: HBM rate of the Multinomial sweep's access pattern without any compute.
// X is N x D Float32 row-major (row = point).  A wave owns 64 points (4 groups of 16) and walks the features in steps of
// `STEP` floats: lane (i, g) loads STEP/4 floats of point i of each group (16-byte pieces, STEP*4 bytes contiguous per point).
//   STEP = 32  -> 128 B per point per step (what mult_sweep_bf16_kernel does), 64  -> 256 B, 128 -> 512 B.
// Also a plain linear copy-read for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int STEP, int AHEAD>
__global__ __launch_bounds__(256) void tile_stream(const float *X, int64_t n, int D, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ci = lane & 15, g = lane >> 4;
    constexpr int PIECES = STEP / 16;          // 16-byte pieces per lane per point per step (4 row groups share a point's STEP floats)
    f32x4 acc = {0, 0, 0, 0};
    const int64_t ntiles = n / 256;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const float *xp[4];
        for (int m = 0; m < 4; ++m) xp[m] = X + (tile * 256 + wave * 64 + 16 * m + ci) * (int64_t)D + (STEP / 4) * g;
        const int nsteps = D / STEP;
        f32x4 buf[AHEAD + 1][4][PIECES];
        for (int a = 0; a < AHEAD; ++a)
            for (int m = 0; m < 4; ++m)
                for (int p = 0; p < PIECES; ++p) buf[a][m][p] = *reinterpret_cast<const f32x4 *>(xp[m] + a * STEP + 4 * p);
        for (int s = 0; s < nsteps; ++s) {
            const int sl = s + AHEAD < nsteps ? s + AHEAD : nsteps - 1;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int p = 0; p < PIECES; ++p) buf[AHEAD][m][p] = *reinterpret_cast<const f32x4 *>(xp[m] + sl * STEP + 4 * p);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int p = 0; p < PIECES; ++p) acc += buf[0][m][p];
#pragma unroll
            for (int a = 0; a < AHEAD; ++a)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int p = 0; p < PIECES; ++p) buf[a][m][p] = buf[a + 1][m][p];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[threadIdx.x] = acc.x;
}

__global__ __launch_bounds__(256) void linear_read(const f32x4 *X, int64_t n4, float *sink) {
    f32x4 acc = {0, 0, 0, 0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) acc += X[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[threadIdx.x] = acc.x;
}

template <class F> float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    const int64_t n = 1 << 20; const int D = 1024;       // 4 GiB, rows 4 KiB (line aligned; the real D = 1000 is not)
    float *X, *sink; hipMalloc(&X, sizeof(float) * n * D); hipMalloc(&sink, 4096); hipMemset(X, 0, sizeof(float) * n * D);
    const double gb = sizeof(float) * (double)n * D / 1e9;
    float ms = time_ms([&] { hipLaunchKernelGGL(linear_read, dim3(256 * 8), dim3(256), 0, 0, (const f32x4 *)X, n * D / 4, sink); });
    printf("linear read                      %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
#define RUN(STEP, AHEAD, GRID) ms = time_ms([&] { hipLaunchKernelGGL((tile_stream<STEP, AHEAD>), dim3(GRID), dim3(256), 0, 0, X, n, D, sink); }); \
    printf("tile stream STEP=%3d AHEAD=%d grid=%4d  %.3f ms  %.0f GB/s\n", STEP, AHEAD, GRID, ms, gb / ms * 1e3);
    RUN(32, 1, 512) RUN(32, 2, 512) RUN(32, 4, 512) RUN(32, 2, 1024) RUN(32, 4, 1024) RUN(32, 4, 2048)
    RUN(64, 1, 512) RUN(64, 2, 512) RUN(64, 2, 1024) RUN(128, 1, 512) RUN(128, 2, 1024)
    return 0;
}
