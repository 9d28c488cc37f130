// Microbenchmark: how fast do N Int32 values stream through one-wave workgroups of 2048 values each (the access pattern of
// hist_kernel / scatter_kernel), against 256-thread workgroups and a persistent grid-stride kernel?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void tile_wave(const int4 *__restrict__ src, int *__restrict__ out) {
    const int4 *p = src + (size_t)blockIdx.x * 512;
    int4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[i * 64 + threadIdx.x];
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 123456789) out[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void tile_wg(const int4 *__restrict__ src, int *__restrict__ out, int ntiles) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int4 *p = src + (size_t)tile * 512;
    int4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[i * 64 + (threadIdx.x & 63)];
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 123456789) out[tile] = s;
}
__global__ __launch_bounds__(256) void persistent(const int4 *__restrict__ src, int *__restrict__ out, size_t n4) {
    int s = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256 * 4) {
        int4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t j = i + (size_t)u * gridDim.x * 256; v[u] = j < n4 ? src[j] : make_int4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 123456789) out[blockIdx.x] = s;
}
int main() {
    const size_t n = 10000000, nt = (n + 2047) / 2048, n4 = nt * 512;
    int4 *src; int *out;
    hipMalloc(&src, n4 * 16); hipMalloc(&out, 4 * (nt + 4096));
    hipMemset(src, 1, n4 * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s %7.2f us per launch  %6.2f TB/s\n", name, 1e3 * ms / 20, n4 * 16.0 / (ms / 20 * 1e-3) / 1e12);
    };
    timeit("one wave per 2048-value tile", [&] { hipLaunchKernelGGL(tile_wave, dim3(nt), dim3(64), 0, 0, src, out); });
    timeit("four tiles per workgroup", [&] { hipLaunchKernelGGL(tile_wg, dim3((nt + 3) / 4), dim3(256), 0, 0, src, out, (int)nt); });
    for (int g : {512, 1024, 2048, 4096})
        { char nm[64]; snprintf(nm, 64, "persistent, %d workgroups", g); timeit(nm, [&] { hipLaunchKernelGGL(persistent, dim3(g), dim3(256), 0, 0, src, out, n4); }); }
    return 0;
}
