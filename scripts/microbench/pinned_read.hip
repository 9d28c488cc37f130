// Dev microbenchmark: how fast does the CPU read pinned host memory that a kernel has just written?  (default / non-coherent /
// write-combined allocations, one thread)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <time.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void fill(double *dst, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v + i;
}
int main() {
    const size_t n = (17u << 20) / 8;
    const unsigned flags[3] = {hipHostMallocDefault, hipHostMallocNonCoherent, hipHostMallocWriteCombined};
    const char *names[3] = {"default", "non-coherent", "write-combined"};
    std::vector<double> dstv(n);
    for (int f = 0; f < 3; ++f) {
        double *h = nullptr;
        if (hipHostMalloc((void **)&h, n * 8, flags[f]) != hipSuccess) { printf("%s: alloc failed\n", names[f]); continue; }
        double best_sum = 1e9, best_cpy = 1e9, kern = 1e9, s = 0;
        for (int rep = 0; rep < 5; ++rep) {
            double t0 = now();
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, h, n, (double)rep);
            hipDeviceSynchronize();
            kern = std::min(kern, now() - t0);
            t0 = now();
            memcpy(dstv.data(), h, n * 8);
            best_cpy = std::min(best_cpy, now() - t0);
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, h, n, (double)rep + 0.5);
            hipDeviceSynchronize();
            t0 = now();
            s = 0; for (size_t i = 0; i < n; ++i) s += h[i];
            best_sum = std::min(best_sum, now() - t0);
        }
        printf("%-15s kernel write %.3f ms (%.1f GB/s)  memcpy out %.3f ms (%.1f GB/s)  scalar sum %.3f ms (%.1f GB/s)  [%g]\n", names[f],
               kern * 1e3, n * 8 / kern / 1e9, best_cpy * 1e3, n * 8 / best_cpy / 1e9, best_sum * 1e3, n * 8 / best_sum / 1e9, s);
        hipHostFree(h);
    }
    return 0;
}
