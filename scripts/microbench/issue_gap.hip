// Microbenchmark (round 5): two waves on one SIMD.  Wave A streams v_mfma_f32_16x16x4_f32 (four independent accumulators) with a pause of
// s_nop cycles behind every matrix instruction; wave B runs a VALU stream.  issue_prio.hip showed that B gets ONE vector instruction per
// matrix instruction of A whatever the priorities (A's next, ready MFMA seems to hold the vector issue port until the matrix pipe frees).
// Question: if A does not present its next MFMA at once, does B get the free issue cycles -- and what does A lose?
//   hipcc --offload-arch=gfx950 -O3 -o issue_gap.bin issue_gap.hip && ./issue_gap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NOPA, int NOPB, int FILL>
__global__ __launch_bounds__(512) void probe(int bmode, int iters, unsigned long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0, t1;
    if (wave >= 4) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = lane * 0.001f, y = 1.0f, f0 = lane, f1 = lane + 1.f;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define ONE(acc)                                                                       \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);                \
        if (FILL >= 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f0));            \
        if (FILL >= 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f1));            \
        if (NOPA > 0) asm volatile("s_nop %0" ::"n"(NOPA - 1));                        \
        if (NOPB > 0) asm volatile("s_nop %0" ::"n"(NOPB - 1));
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { ONE(a0) ONE(a1) ONE(a2) ONE(a3) }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1;
        return;
    }
    if (bmode < 0) return;
    for (int i = 0; i < 20; ++i) __builtin_amdgcn_s_sleep(10);
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int n = iters / 2;      // 32 iters instructions
    if (bmode == 0) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v0));
        }
    } else {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)n * 64; }
    sink[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3;
}

template <int NOPA, int NOPB, int FILL>
static void run(unsigned long long *d_out, float *d_sink, std::vector<unsigned long long> &h) {
    const int grid = 256, iters = 200;
    for (int bmode = -1; bmode < 2; ++bmode) {
        hipMemset(d_out, 0, sizeof(unsigned long long) * grid * 16);
        hipLaunchKernelGGL((probe<NOPA, NOPB, FILL>), dim3(grid), dim3(512), 0, 0, bmode, iters, d_out, d_sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * grid * 16, hipMemcpyDeviceToHost);
        double sv = 0, sm = 0, nv = 0; int cv = 0, cm = 0;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 8; ++w) {
                const unsigned long long t = h[(b * 8 + w) * 2];
                if (!t) continue;
                if (w < 4) { sv += (double)t; nv += (double)h[(b * 8 + w) * 2 + 1]; ++cv; } else { sm += (double)t; ++cm; }
            }
        printf("A: MFMA + %d own v_fma + s_nop %2d cycles | B: %-26s : A %6.2f cycles per MFMA", FILL, NOPA + NOPB,
               bmode < 0 ? "(absent)" : bmode == 0 ? "dependent v_fma chain" : "4 independent v_fma chains", sm / cm / (iters * 64.0));
        if (cv) printf(" | B %6.2f cycles per instruction", sv / nv);
        printf("\n");
    }
}

int main() {
    unsigned long long *d_out; float *d_sink;
    hipMalloc(&d_out, sizeof(unsigned long long) * 256 * 16);
    hipMalloc(&d_sink, sizeof(float) * 256 * 512);
    std::vector<unsigned long long> h(256 * 16);
    run<0, 0, 0>(d_out, d_sink, h);
    run<4, 0, 0>(d_out, d_sink, h);
    run<8, 0, 0>(d_out, d_sink, h);
    run<12, 0, 0>(d_out, d_sink, h);
    run<16, 0, 0>(d_out, d_sink, h);
    run<16, 4, 0>(d_out, d_sink, h);
    run<16, 8, 0>(d_out, d_sink, h);
    run<16, 12, 0>(d_out, d_sink, h);
    run<0, 0, 1>(d_out, d_sink, h);
    run<0, 0, 2>(d_out, d_sink, h);
    run<8, 0, 2>(d_out, d_sink, h);
    run<16, 0, 2>(d_out, d_sink, h);
    return 0;
}
