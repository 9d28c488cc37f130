// How fast does HBM deliver 1 KiB rows visited in a random order, (a) 128 bytes per visit, eight visits spread over the kernel (what a wave of
// mult_sweep_u8_kernel does: one k-step of 128 features per visit), (b) the whole row in one visit, (c) rows in storage order?
//   hipcc --offload-arch=gfx950 -O3 -o gather_rows gather_rows.hip && ./gather_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: per tile of 256 rows (64 per wave), for ks in 0..7: each wave loads 128 B of its 64 rows (8 full-line loads), then "works" (spin) a little
// MODE 1: per tile, each wave loads its 64 rows completely, 8 rows (8 KiB) per step
template <int MODE>
__global__ __launch_bounds__(256, 2) void gather(const unsigned char *__restrict__ X, const int *__restrict__ order, long n, unsigned *out, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ntiles = n / 256;
    u32x4 acc = {0, 0, 0, 0};
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long i0 = tile * 256 + wave * 64;
        const int myp = order[i0 + lane];
        if (MODE == 0) {
            int rel[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) rel[j] = __shfl(myp, 8 * j + (lane >> 3));
            for (int ks = 0; ks < 8; ++ks) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4 *>(X + (long)rel[j] * 1024 + 128 * ks + 16 * (lane & 7));
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += v[j];
                for (int s = 0; s < spin; ++s) asm volatile("s_nop 15");
            }
        } else {
            for (int st = 0; st < 8; ++st) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = __shfl(myp, 8 * st + j);                 // one row per instruction: 64 lanes x 16 B = the whole KiB
                    v[j] = *reinterpret_cast<const u32x4 *>(X + (long)r * 1024 + 16 * lane);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += v[j];
                for (int s = 0; s < spin; ++s) asm volatile("s_nop 15");
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

int main() {
    const long n = 1000000 / 256 * 256;
    unsigned char *X; int *order; unsigned *out;
    hipMalloc(&X, n * 1024); hipMalloc(&order, n * 4); hipMalloc(&out, 4);
    hipMemset(X, 1, n * 1024);
    std::vector<int> h(n);
    for (long i = 0; i < n; ++i) h[i] = (int)i;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ord = 0; ord < 3; ++ord) {
        if (ord == 1) { std::mt19937 g(1); std::shuffle(h.begin(), h.end(), g); }
        if (ord == 2) {   // 32 clusters, each an increasing subsequence of the storage order (what sorting by label gives)
            std::mt19937 g(2); std::vector<int> lab(n); for (auto &l : lab) l = g() & 31;
            std::vector<int> idx(n); for (long i = 0; i < n; ++i) idx[i] = (int)i;
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lab[a] < lab[b]; });
            h = idx;
        }
        hipMemcpy(order, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode)
            for (int spin : {0, 8, 32}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) gather<0><<<512, 256>>>(X, order, n, out, spin); else gather<1><<<512, 256>>>(X, order, n, out, spin);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                printf("order %s  mode %s  spin %2d: %.3f ms  %.2f TB/s\n", ord == 0 ? "storage" : ord == 1 ? "random " : "by-label", mode == 0 ? "128B x 8 visits" : "whole rows     ", spin, best, n * 1024.0 / best / 1e9);
            }
    }
    return 0;
}
