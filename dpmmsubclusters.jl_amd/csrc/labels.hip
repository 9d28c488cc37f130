// labels.hip -- integer label bookkeeping on the device-resident bin vector.
// bins[i] = 2*(label-1) + (sub_label-1).  All kernels are elementwise, HBM-bound
// (4 B read + 4 B write per point) and exact.
//
// Reference functions replaced (paths relative to the reference checkout):
//   rand(1:init_clusters), rand(1:2)      src/dp-parallel-sampling.jl:49-50
//   split_cluster_local_worker!           src/local_clusters_actions.jl:265-278
//   merge_clusters_worker!                src/local_clusters_actions.jl:293-304
//   remove_empty_clusters_worker!         src/local_clusters_actions.jl:446-455
//   reset_bad_clusters_worker!            src/local_clusters_actions.jl:481-488 (+ :474-479, :257-261)
#include "dpmm_device.h"
#include "dpmm_kernels.h"

namespace dpmm {

static inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

__global__ void init_labels_kernel(int32_t *bins, int64_t n, int64_t first, int init_clusters, uint64_t seed, uint32_t epoch) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const Philox4 r = philox4x32_10(seed, (uint64_t)(first + i), epoch, STREAM_INIT);
        const int z = (int)(((uint64_t)r.v[0] * (uint64_t)init_clusters) >> 32);
        bins[i] = 2 * z + (int)(r.v[1] & 1u);
    }
}

__global__ void bins_from_i64_kernel(int32_t *bins, const int64_t *labels, const int64_t *sub, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int b = bins[i];
        int z = b >> 1, s = b & 1;
        if (labels) z = (int)(labels[i] - 1);
        if (sub) s = (int)(sub[i] - 1);
        bins[i] = 2 * z + s;
    }
}

__global__ void bins_to_i64_kernel(const int32_t *bins, int64_t *labels, int64_t *sub, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = bins[i];
        if (labels) labels[i] = (int64_t)(b >> 1) + 1;
        if (sub) sub[i] = (int64_t)(b & 1) + 1;
    }
}

// pair by pair, in order: label==idx & sub==2 -> new_idx ; every point that had label idx gets rand(1:2)
__global__ void split_kernel(int32_t *bins, int64_t n, int64_t first, const int32_t *pairs, int m, uint64_t seed, uint32_t epoch) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = bins[i];
        int z = b >> 1, s = b & 1;
        bool touched = false;
        for (int j = 0; j < m; ++j) {
            if (z == pairs[j]) {
                if (s == 1) z = pairs[m + j];
                if (!touched) {
                    touched = true;
                }
                const Philox4 r = philox4x32_10(seed, (uint64_t)(first + i), epoch, STREAM_SPLIT);
                s = (int)(r.v[0] & 1u);
            }
        }
        if (touched) bins[i] = 2 * z + s;
    }
}

__global__ void merge_kernel(int32_t *bins, int64_t n, const int32_t *pairs, int m) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = bins[i];
        int z = b >> 1, s = b & 1;
        for (int j = 0; j < m; ++j) {
            if (z == pairs[j]) s = 0;
            if (z == pairs[m + j]) { s = 1; z = pairs[j]; }
        }
        const int nb = 2 * z + s;
        if (nb != b) bins[i] = nb;
    }
}

__global__ void remap_kernel(int32_t *bins, int64_t n, const int32_t *map) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = bins[i];
        const int z = map[b >> 1];
        const int nb = 2 * z + (b & 1);
        if (nb != b) bins[i] = nb;
    }
}

__global__ void reset_sub_kernel(int32_t *bins, int64_t n, int64_t first, const int32_t *idx, int m, uint64_t seed, uint32_t epoch) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = bins[i];
        const int z = b >> 1;
        bool hit = (idx == nullptr);
        for (int j = 0; j < m && !hit; ++j) hit = (z == idx[j]);
        if (hit) {
            const Philox4 r = philox4x32_10(seed, (uint64_t)(first + i), epoch, STREAM_RESET);
            bins[i] = 2 * z + (int)(r.v[0] & 1u);
        }
    }
}

// Per-step host<->device transfers go through pinned (GPU-addressable) staging and this kernel instead of
// hipMemcpyAsync: the copy-engine path stalled the stream for tens of ms every dozen steps on the test box.
__global__ void copy_words_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, int64_t nwords) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
hipError_t launch_copy_bytes(void *dst, const void *src, size_t bytes, hipStream_t s) {
    const int64_t nw = (int64_t)((bytes + 3) / 4);
    if (nw == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_words_kernel, dim3(grid_for(nw)), dim3(256), 0, s, (uint32_t *)dst, (const uint32_t *)src, nw);
    return hipGetLastError();
}

// counts[k * n_gt + g] += 1 over the shard; per-workgroup LDS privatisation when the table is small
__global__ void contingency_kernel(const int32_t *__restrict__ bins, const int32_t *__restrict__ gt, int64_t n, int K, int n_gt,
                                   unsigned long long *__restrict__ counts) {
    extern __shared__ unsigned int tab[];
    const int cells = K * n_gt;
    const bool use_lds = cells <= 8192;
    if (use_lds) {
        for (int c = threadIdx.x; c < cells; c += blockDim.x) tab[c] = 0u;
        __syncthreads();
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = bins[i] >> 1, g = gt[i];
        if ((unsigned)k < (unsigned)K && (unsigned)g < (unsigned)n_gt) {
            if (use_lds) atomicAdd(&tab[k * n_gt + g], 1u);
            else atomicAdd(&counts[(size_t)k * n_gt + g], 1ull);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int c = threadIdx.x; c < cells; c += blockDim.x)
            if (tab[c]) atomicAdd(&counts[c], (unsigned long long)tab[c]);
    }
}
hipError_t launch_contingency(const int32_t *bins, const int32_t *gt, int64_t n, int K, int n_gt, unsigned long long *counts, hipStream_t s) {
    const int cells = K * n_gt;
    const size_t lds = cells <= 8192 ? sizeof(unsigned) * cells : 0;
    hipLaunchKernelGGL(contingency_kernel, dim3(grid_for(n)), dim3(256), lds, s, bins, gt, n, K, n_gt, counts);
    return hipGetLastError();
}
__global__ void i64_to_i32_kernel(int32_t *dst, const int64_t *src, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (int32_t)src[i];
}
hipError_t launch_i64_to_i32(int32_t *dst, const int64_t *src, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(i64_to_i32_kernel, dim3(grid_for(n)), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}

// .npy ingestion (utils.jl:5-14): rows of Float32 / Float64 samples -> the ctx layout (Float32, leading dimension ldx),
// NaN -> 0.  Elementwise, HBM-bound.
template <typename T>
__global__ void ingest_rows_kernel(float *__restrict__ dst, int64_t ldx, const T *__restrict__ src, int64_t ld, int64_t rows, int D,
                                   int nan_to_zero) {
    const int64_t total = rows * D;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / D;
        const int d = (int)(e - r * D);
        float v = (float)src[r * ld + d];
        if (nan_to_zero && v != v) v = 0.f;
        dst[r * ldx + d] = v;
    }
}
hipError_t launch_ingest_rows(float *dst, int64_t ldx, const void *src, int is_f64, int64_t ld, int64_t rows, int D, int nan_to_zero,
                              hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (is_f64)
        hipLaunchKernelGGL((ingest_rows_kernel<double>), dim3(grid_for(rows * D)), dim3(256), 0, s, dst, ldx, (const double *)src, ld, rows, D, nan_to_zero);
    else
        hipLaunchKernelGGL((ingest_rows_kernel<float>), dim3(grid_for(rows * D)), dim3(256), 0, s, dst, ldx, (const float *)src, ld, rows, D, nan_to_zero);
    return hipGetLastError();
}

hipError_t launch_init_labels(int32_t *bins, int64_t n, int64_t first, int init_clusters, uint64_t seed, uint32_t epoch, hipStream_t s) {
    hipLaunchKernelGGL(init_labels_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, first, init_clusters, seed, epoch);
    return hipGetLastError();
}
hipError_t launch_bins_from_i64(int32_t *bins, const int64_t *labels, const int64_t *sub, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(bins_from_i64_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, labels, sub, n);
    return hipGetLastError();
}
hipError_t launch_bins_to_i64(const int32_t *bins, int64_t *labels, int64_t *sub, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(bins_to_i64_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, labels, sub, n);
    return hipGetLastError();
}
hipError_t launch_split(int32_t *bins, int64_t n, int64_t first, const int32_t *pairs, int m, uint64_t seed, uint32_t epoch, hipStream_t s) {
    hipLaunchKernelGGL(split_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, first, pairs, m, seed, epoch);
    return hipGetLastError();
}
hipError_t launch_merge(int32_t *bins, int64_t n, const int32_t *pairs, int m, hipStream_t s) {
    hipLaunchKernelGGL(merge_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, pairs, m);
    return hipGetLastError();
}
hipError_t launch_remap(int32_t *bins, int64_t n, const int32_t *map, hipStream_t s) {
    hipLaunchKernelGGL(remap_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, map);
    return hipGetLastError();
}
hipError_t launch_reset_sub(int32_t *bins, int64_t n, int64_t first, const int32_t *idx, int m, uint64_t seed, uint32_t epoch, hipStream_t s) {
    hipLaunchKernelGGL(reset_sub_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, first, idx, m, seed, epoch);
    return hipGetLastError();
}

}  // namespace dpmm
