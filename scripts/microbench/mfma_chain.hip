// Microbenchmark (round 6): what does a DEPENDENT chain of v_mfma_f32_16x16x32_bf16 cost against the same count spread over
// independent accumulators?  The three-plane evaluation (niw_b3.h b3_eval) issues, per point group, 12 MFMAs into one accumulator;
// the compiler keeps the four point groups' chains mostly one after another.  WAVES = waves per SIMD issuing the pattern.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain.bin mfma_chain.hip && ./mfma_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long *out, float *sink) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane + i); b[i] = (short)(0x3f00 + lane * 2 + i); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (NACC == 1) {
            // four chains of 12, one after another (the order the compiler emits today)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int u = 0; u < 12; ++u)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(b));
        } else if constexpr (NACC == 2) {
#pragma unroll
            for (int n = 0; n < 4; n += 2)
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(b));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[n + 1]) : "v"(a), "v"(b));
                }
        } else {
#pragma unroll
            for (int u = 0; u < 12; ++u)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int NACC>
static void run(int waves_per_simd, const char *name) {
    const int iters = 2000, threads = 256 * waves_per_simd;
    unsigned long long *out; float *sink;
    hipMalloc(&out, 64 * sizeof(unsigned long long)); hipMalloc(&sink, threads * sizeof(float));
    probe<NACC><<<1, threads>>>(iters, out, sink);
    probe<NACC><<<1, threads>>>(iters, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(threads / 64);
    hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz; the core at ~2.4 GHz (clock_probe.hip): report ticks per MFMA per wave and per SIMD
    double t = (double)h[0] / (iters * 48.0);
    printf("%-36s waves/SIMD %d: %.3f memtime ticks per MFMA per wave, %.3f per SIMD-MFMA\n", name, waves_per_simd, t, t / waves_per_simd);
    hipFree(out); hipFree(sink);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<1>(w, "one accumulator, chains in sequence");
        run<2>(w, "two accumulators interleaved");
        run<4>(w, "four accumulators interleaved");
    }
    return 0;
}
