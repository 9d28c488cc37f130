// Microbenchmark: how fast does a wave issue VALU / SALU work while ANOTHER wave on the same SIMD streams MFMAs?
//   hipcc --offload-arch=gfx950 -O3 -o issue_overlap issue_overlap.hip && ./issue_overlap
// One workgroup of 512 threads per CU: waves 0-3 land on SIMD 0-3 and so do waves 4-7 (checked through HW_ID).
// Waves 0-3 run the probe (a chain of VALU or SALU instructions, timed with s_memtime); waves 4-7 either idle or issue
// back-to-back v_mfma_f32_16x16x4_f32 (independent accumulators) for the whole time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(int mode, int mfma_on, int iters, unsigned long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    if (wave >= 4) {
        if (!mfma_on) return;
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = lane * 0.001f, y = 1.0f;
        for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        }
        sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        return;
    }
    // let the MFMA waves get going
    for (int i = 0; i < 50; ++i) __builtin_amdgcn_s_sleep(10);
    unsigned long long t0, t1;
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    int s0 = 1, s1 = 2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (mode == 0) {            // dependent VALU chain (1 chain)
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v0));
        }
    } else if (mode == 1) {     // 4 independent VALU chains
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
    } else if (mode == 2) {     // SALU chain
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
        }
    } else if (mode == 4) {     // 4 independent packed-f32 chains (v_pk_fma_f32: two f32 per lane per instruction)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 p0 = {v0, v1}, p1 = {v1, v2}, p2 = {v2, v3}, p3 = {v3, v0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\tv_pk_fma_f32 %3, %3, %3, %3"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        }
        v0 = p0.x + p1.y + p2.x + p3.y;
    } else if (mode == 5) {     // LDS reads (ds_read_b128, independent)
        extern __shared__ float lds[];
        f32x4 acc = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) acc += *reinterpret_cast<volatile f32x4 *>(lds + ((lane * 4 + u * 256) & 4095));
        }
        v0 = acc.x + acc.y;
    } else {                    // alternating VALU / SALU
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %0, %0, %0\n\ts_add_u32 %1, %1, %2" : "+v"(v0), "+s"(s0) : "s"(s1) : "scc");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) {
        out[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 4 + wave) * 2 + 1] = hwid;
    }
    sink[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + s0;
}

int main() {
    const int grid = 256, iters = 200;
    unsigned long long *d_out; float *d_sink;
    hipMalloc(&d_out, sizeof(unsigned long long) * grid * 8);
    hipMalloc(&d_sink, sizeof(float) * grid * 512);
    std::vector<unsigned long long> h(grid * 8);
    const char *names[7] = {"dependent VALU chain", "4 independent VALU chains", "SALU chain", "VALU/SALU alternating (per pair /2)", "4 independent v_pk_fma_f32 chains", "LDS ds_read_b128 stream", "VALU/SALU alternating"};
    const int order[6] = {0, 1, 4, 2, 6, 5};
    for (int oi = 0; oi < 6; ++oi) { const int mode = order[oi];
        for (int on = 0; on < 2; ++on) {
            hipMemset(d_out, 0, sizeof(unsigned long long) * grid * 8);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 16384, 0, mode == 6 ? 3 : mode, on, iters, d_out, d_sink);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * grid * 8, hipMemcpyDeviceToHost);
            double sum = 0; int cnt = 0;
            for (int i = 0; i < grid * 4; ++i) if (h[2 * i]) { sum += (double)h[2 * i]; ++cnt; }
            const double per_instr = sum / cnt / (iters * 64.0);
            printf("%-28s MFMA neighbour %s: %.2f cycles per instruction (simd id of wave0: %llu)\n", names[mode], on ? "ON " : "off", per_instr,
                   (h[1] >> 4) & 3ull);
        }
    }
    return 0;
}
