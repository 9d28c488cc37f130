// Microbenchmark (round 5): which FP32-input matrix instruction shapes leave the SIMD's vector issue port free while they execute?
// One wave per SIMD streams one shape (4 independent accumulators) with FILL own independent v_fma_f32 behind every matrix instruction;
// optionally a second wave on the same SIMD runs 4 independent v_fma chains.  v_mfma_f32_16x16x4_f32 (the sweep kernel's instruction)
// hides NOTHING: every own filler adds its full issue time, and the other wave gets one instruction per one or two MFMAs (issue_gap.hip).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shapes_issue.bin mfma_shapes_issue.hip && ./mfma_shapes_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int FILL>
__global__ __launch_bounds__(512) void probe(int bmode, int iters, unsigned long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0, t1;
    if (wave >= 4) {
        float x = lane * 0.001f, y = 1.0f, f0 = lane, f1 = lane + 1.f, f2 = lane + 2.f, f3 = lane + 3.f;
        float res = 0.f;
#define FILLERS()                                                                      \
        if (FILL >= 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f0));            \
        if (FILL >= 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f1));            \
        if (FILL >= 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f2));            \
        if (FILL >= 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f3));
        if constexpr (SHAPE == 0) {            // 16x16x4 f32: 32 cycles, 2048 flop
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); FILLERS()
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0); FILLERS()
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); FILLERS()
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0); FILLERS()
                }
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            res = a0[0] + a1[1] + a2[2] + a3[3];
        } else if constexpr (SHAPE == 1) {     // 32x32x2 f32: 64 cycles, 4096 flop
            f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); FILLERS()
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0); FILLERS()
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); FILLERS()
                    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0); FILLERS()
                }
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            res = a0[0] + a1[1] + a2[2] + a3[3];
        } else if constexpr (SHAPE == 2) {     // 16x16x1 x 4 blocks f32: 32 cycles, 2048 flop
            f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a0, 0, 0, 0); FILLERS()
                    a1 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a1, 0, 0, 0); FILLERS()
                    a2 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a2, 0, 0, 0); FILLERS()
                    a3 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a3, 0, 0, 0); FILLERS()
                }
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            res = a0[0] + a1[1] + a2[2] + a3[3];
        } else if constexpr (SHAPE == 3) {     // 4x4x1 x 16 blocks f32: 8 cycles, 512 flop
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0); FILLERS()
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0); FILLERS()
                    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0); FILLERS()
                    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0); FILLERS()
                }
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            res = a0[0] + a1[1] + a2[2] + a3[3];
        } else if constexpr (SHAPE == 4) {     // 32x32x1 x 2 blocks f32: 64 cycles, 4096 flop
            f32x32 a0 = {}, a1 = {};
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x1f32(x, y, a0, 0, 0, 0); FILLERS()
                    a1 = __builtin_amdgcn_mfma_f32_32x32x1f32(x, y, a1, 0, 0, 0); FILLERS()
                }
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            res = a0[0] + a1[1];
        } else {                               // 16x16x32 bf16 (reference: the shape the guide's co-issue numbers are for): 16 cycles
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            bf16x8 p = {(short)lane, 1, 2, 3, 4, 5, 6, 7}, q = {1, 1, 1, 1, 1, 1, 1, (short)lane};
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a0, 0, 0, 0); FILLERS()
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a1, 0, 0, 0); FILLERS()
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a2, 0, 0, 0); FILLERS()
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a3, 0, 0, 0); FILLERS()
                }
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            res = a0[0] + a1[1] + a2[2] + a3[3];
        }
        if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        sink[blockIdx.x * 512 + threadIdx.x] = res + f0 + f1 + f2 + f3;
        return;
    }
    if (bmode < 0) return;
    for (int i = 0; i < 20; ++i) __builtin_amdgcn_s_sleep(10);
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int n = iters / 4;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)n * 64; }
    sink[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3;
}

static const char *names[6] = {"16x16x4_f32 ", "32x32x2_f32 ", "16x16x1_4B  ", "4x4x1_16B   ", "32x32x1_2B  ", "16x16x32bf16"};
template <int SHAPE, int FILL>
static void run(unsigned long long *d_out, float *d_sink, std::vector<unsigned long long> &h) {
    const int grid = 256, iters = 200;
    for (int bmode = -1; bmode < 1; ++bmode) {
        hipMemset(d_out, 0, sizeof(unsigned long long) * grid * 16);
        hipLaunchKernelGGL((probe<SHAPE, FILL>), dim3(grid), dim3(512), 0, 0, bmode, iters, d_out, d_sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * grid * 16, hipMemcpyDeviceToHost);
        double sv = 0, sm = 0, nv = 0; int cv = 0, cm = 0;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 8; ++w) {
                const unsigned long long t = h[(b * 8 + w) * 2];
                if (!t) continue;
                if (w < 4) { sv += (double)t; nv += (double)h[(b * 8 + w) * 2 + 1]; ++cv; } else { sm += (double)t; ++cm; }
            }
        printf("%s + %d own v_fma | other wave %-22s : %7.2f cycles per MFMA", names[SHAPE], FILL, bmode < 0 ? "absent" : "4 indep. v_fma chains", sm / cm / (iters * 64.0));
        if (cv) printf(" | other wave %6.2f cycles per instruction", sv / nv);
        printf("\n");
    }
}
template <int SHAPE>
static void runs(unsigned long long *d_out, float *d_sink, std::vector<unsigned long long> &h) {
    run<SHAPE, 0>(d_out, d_sink, h); run<SHAPE, 1>(d_out, d_sink, h); run<SHAPE, 2>(d_out, d_sink, h); run<SHAPE, 4>(d_out, d_sink, h);
}
int main() {
    unsigned long long *d_out; float *d_sink;
    hipMalloc(&d_out, sizeof(unsigned long long) * 256 * 16);
    hipMalloc(&d_sink, sizeof(float) * 256 * 512);
    std::vector<unsigned long long> h(256 * 16);
    runs<0>(d_out, d_sink, h); runs<1>(d_out, d_sink, h); runs<2>(d_out, d_sink, h); runs<3>(d_out, d_sink, h); runs<4>(d_out, d_sink, h); runs<5>(d_out, d_sink, h);
    return 0;
}
