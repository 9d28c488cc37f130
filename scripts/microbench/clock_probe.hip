// Dev microbenchmark: what does s_memtime count, and at what clock does the chip run a kernel?
// Every wave issues NM dependent v_mfma_f32_16x16x4_f32 (32 shader cycles each, measured rate of mfma_f32_peak.hip), optionally with
// LDS reads + VALU work in between; we print wall time (events), s_memtime ticks per MFMA and the implied rates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long *out, float *sink) {
    __shared__ float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
    float x = lane * 0.001f, y = 1.0f, v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE >= 1) {
                const f32x4 f = *reinterpret_cast<const f32x4 *>(lds + ((it * 16 + u) & 63) * 256 + lane * 4);
                x = f.x; y = f.y;
            }
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            if (MODE >= 2) {
#pragma unroll
                for (int f = 0; f < 4; ++f) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[f]));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = a0[0] + a1[1];
    for (int i = 0; i < 8; ++i) s += v[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int grid) {
    const int iters = 4000;
    unsigned long long *d_out; float *d_sink;
    (void)hipMalloc(&d_out, 8 * grid * 4); (void)hipMalloc(&d_sink, 4 * grid * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, iters, d_out, d_sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, iters, d_out, d_sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), d_out, 8 * grid * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (auto c : h) sum += (double)c;
    const double ticks = sum / h.size(), nm = iters * 32.0;
    printf("%-26s grid %4d: %.3f ms wall, %.1f s_memtime ticks per MFMA, s_memtime rate %.0f MHz, MFMA rate per wave %.1f ns (32 cycles at %.2f GHz if back to back)\n",
           name, grid, ms, ticks / nm, ticks / (ms * 1e3), ms * 1e6 / nm, 32.0 / (ms * 1e6 / nm));
    hipFree(d_out); hipFree(d_sink);
}
int main() {
    for (int grid : {256, 512}) {
        run<0>("MFMA only", grid);
        run<1>("MFMA + LDS reads", grid);
        run<2>("MFMA + LDS + 2 VALU/MFMA", grid);
    }
    return 0;
}
