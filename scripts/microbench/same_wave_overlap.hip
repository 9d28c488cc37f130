// Dev microbenchmark: how many VALU "fillers" does ONE wave hide behind its OWN v_mfma_f32_16x16x4_f32 stream (one wave per SIMD)?
// Decides whether a software-pipelined sweep kernel (tile B's matrix phase interleaved with tile A's screen / draw VALU in the same
// wave) can approach the matrix-pipe bound.  Fillers: plain v_fma_f32, packed v_pk_fma_f32, and v_fma_f32 + an SGPR op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NF, int KIND>   // NF fillers after every MFMA; KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_fma_f32 + s_add
__global__ __launch_bounds__(256, 1) void k(int iters, unsigned long long *out, float *sink) {
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 0.001f, y = 1.0f;
    float v[8]; f32x2 p[8]; int s0 = 1;
    for (int i = 0; i < 8; ++i) { v[i] = lane + i; p[i] = (f32x2){(float)lane, (float)i}; }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            f32x4 &acc = (u & 3) == 0 ? a0 : (u & 3) == 1 ? a1 : (u & 3) == 2 ? a2 : a3;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[f & 7]));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[f & 7]));
                else asm volatile("v_fma_f32 %0, %0, %0, %0\n\ts_add_u32 %1, %1, 1" : "+v"(v[f & 7]), "+s"(s0) : : "scc");
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = a0[0] + a1[1] + a2[2] + a3[3] + s0;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NF, int KIND>
void run(const char *name) {
    const int grid = 256, iters = 500;
    static unsigned long long *d_out = nullptr; static float *d_sink = nullptr;
    if (!d_out) { (void)hipMalloc(&d_out, 8 * grid * 4); (void)hipMalloc(&d_sink, 4 * grid * 256); }
    std::vector<unsigned long long> h(grid * 4);
    hipLaunchKernelGGL((k<NF, KIND>), dim3(grid), dim3(256), 0, 0, iters, d_out, d_sink);
    hipLaunchKernelGGL((k<NF, KIND>), dim3(grid), dim3(256), 0, 0, iters, d_out, d_sink);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_out, 8 * grid * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (auto c : h) sum += (double)c;
    printf("%-14s %d fillers per MFMA: %.1f cycles per MFMA slot (s_memtime ticks x clock ratio not applied)\n", name, NF, sum / h.size() / (iters * 16.0));
}
int main() {
    run<0, 0>("v_fma_f32"); run<2, 0>("v_fma_f32"); run<4, 0>("v_fma_f32"); run<6, 0>("v_fma_f32"); run<8, 0>("v_fma_f32"); run<12, 0>("v_fma_f32");
    run<2, 1>("v_pk_fma_f32"); run<4, 1>("v_pk_fma_f32"); run<6, 1>("v_pk_fma_f32");
    run<2, 2>("fma + s_add"); run<4, 2>("fma + s_add"); run<6, 2>("fma + s_add");
    return 0;
}
