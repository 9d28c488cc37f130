// Microbenchmark (round 5): two waves on one SIMD, one streaming v_mfma_f32_16x16x4_f32 (four independent accumulators), the other a VALU
// stream -- cycles per VALU instruction AND cycles per MFMA, with and without s_setprio on the VALU wave.  Question: can the vector work of
// one wave hide in the 24 free issue cycles of the other wave's 32-cycle matrix instruction if the arbiter is told to prefer it?
//   hipcc --offload-arch=gfx950 -O3 -o issue_prio.bin issue_prio.hip && ./issue_prio.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(int mode, int prio_valu, int prio_mfma, int mfma_on, int iters, unsigned long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0, t1;
    if (wave >= 4) {
        if (!mfma_on) return;
        if (prio_mfma == 1) __builtin_amdgcn_s_setprio(1);
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = lane * 0.001f, y = 1.0f;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        return;
    }
    if (mode < 0) return;                   // MFMA waves alone
    if (prio_valu) __builtin_amdgcn_s_setprio(3);
    if (mode == 2) {            // the sweep kernel's own mix: x - mu subtractions feeding MFMAs of THIS wave (does a wave's own MFMA stream share the pipe fairly?)
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = lane * 0.001f, y = 1.0f;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)iters * 64; }
        sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        return;
    }
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    // the VALU wave runs ~ as long as the MFMA wave: iters * 64 MFMAs * 32 cycles = iters * 2048 cycles
    for (int i = 0; i < 20; ++i) __builtin_amdgcn_s_sleep(10);      // let the MFMA waves get going
    const int n = iters;      // 64 n instructions: well inside the MFMA stream (iters * 2048 cycles)
    if (mode == 0) {            // dependent chain
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v0));
        }
    } else {                    // 4 independent chains
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

int main() {
    const int grid = 256, iters = 100;
    unsigned long long *d_out; float *d_sink;
    hipMalloc(&d_out, sizeof(unsigned long long) * grid * 16);
    hipMalloc(&d_sink, sizeof(float) * grid * 512);
    std::vector<unsigned long long> h(grid * 16);
    const char *mname[4] = {"(none: MFMA alone)", "dependent v_fma chain", "4 independent v_fma chains", "a second MFMA stream"};
    for (int mode = -1; mode < 3; ++mode)
        for (int pv = 0; pv < 2; ++pv)
            for (int pm = 0; pm < 3; ++pm) {
                if (mode < 0 && (pv || pm)) continue;
                const int mfma_on = pm < 2;
                hipMemset(d_out, 0, sizeof(unsigned long long) * grid * 16);
                hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, mode, pv, pm, mfma_on, iters, d_out, d_sink);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * grid * 16, hipMemcpyDeviceToHost);
                double sv = 0, sm = 0, nv = 0; int cv = 0, cm = 0;
                for (int b = 0; b < grid; ++b)
                    for (int w = 0; w < 8; ++w) {
                        const unsigned long long t = h[(b * 8 + w) * 2];
                        if (!t) continue;
                        if (w < 4) { sv += (double)t; nv += (double)h[(b * 8 + w) * 2 + 1]; ++cv; } else { sm += (double)t; ++cm; }
                    }
                printf("other wave: %-28s prio(other=%d, mfma=%d) MFMA wave %s: ", mname[mode + 1], pv ? 3 : 0, pm & 1, mfma_on ? "on " : "off");
                if (cm) printf("%6.2f cycles per MFMA (32 = pipe rate)", sm / cm / (iters * 64.0));
                if (cv) printf(" | %6.2f cycles per instruction of the other wave", sv / nv);
                printf("\n");
            }
    return 0;
}
