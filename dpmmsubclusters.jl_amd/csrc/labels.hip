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

__global__ void init_labels_kernel(int32_t *bins, int64_t n, int64_t first, int init_clusters, int label0, uint64_t seed, uint32_t epoch) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const Philox4 r = philox4x32_10(seed, (uint64_t)(first + i), epoch, STREAM_INIT);
        const int z = label0 + (int)(((uint64_t)r.v[0] * (uint64_t)init_clusters) >> 32);
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

// reset_bad_clusters! without a host round trip: the sub-cluster occupancies (global over all shards) decide on the device
// which clusters are "bad" (an empty sub-cluster, local_clusters_actions.jl:501-516); flags[K] = any.
__global__ void bad_flags_kernel(const int32_t *__restrict__ bin_total, const long long *__restrict__ global_counts, int K,
                                 uint8_t *__restrict__ flags) {
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const long long a = global_counts ? global_counts[2 * k] : (long long)bin_total[2 * k];
        const long long b = global_counts ? global_counts[2 * k + 1] : (long long)bin_total[2 * k + 1];
        const int bad = (a == 0 || b == 0) ? 1 : 0;
        flags[k] = (uint8_t)bad;
        if (bad) atomicOr(&any, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) flags[K] = (uint8_t)any;
}
hipError_t launch_bad_flags(const int32_t *bin_total, const long long *global_counts, int K, uint8_t *flags, hipStream_t s) {
    DPMM_LAUNCH(bad_flags_kernel, dim3(1), dim3(256), 0, s, bin_total, global_counts, K, flags);
    return hipGetLastError();
}
// (the reset itself, with the flag computation folded in, is reset_recount_kernel in suffstats.hip: it re-counts the sort tiles it touches)
__global__ void widen_counts_kernel(const int32_t *__restrict__ src, int stride, long long *__restrict__ dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = (long long)src[(int64_t)i * stride];
}
hipError_t launch_widen_counts(const int32_t *src, int stride, long long *dst, int n, hipStream_t s) {
    DPMM_LAUNCH(widen_counts_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, stride, dst, n);
    return hipGetLastError();
}
// dst[3k+w][0..D) = src[3*slot[k]+w][0..D)  (Multinomial parameter rows from the slot-indexed staging; padding beyond D is zeroed)
__global__ void gather_rows_kernel(float *__restrict__ dst, int64_t ld_dst, const float *__restrict__ src, int64_t ld_src,
                                   const int32_t *__restrict__ slot, int rows, int D) {
    const int64_t total = (int64_t)rows * ld_dst;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e / ld_dst), d = (int)(e - (int64_t)j * ld_dst);
        const size_t sj = slot ? (size_t)(3 * slot[j / 3] + j % 3) : (size_t)j;
        dst[e] = d < D ? src[sj * ld_src + d] : 0.f;
    }
}
hipError_t launch_gather_rows(float *dst, int64_t ld_dst, const float *src, int64_t ld_src, const int32_t *slot, int rows, int D, hipStream_t s) {
    DPMM_LAUNCH(gather_rows_kernel, dim3(grid_for((int64_t)rows * ld_dst)), dim3(256), 0, s, dst, ld_dst, src, ld_src, slot, rows, D);
    return hipGetLastError();
}

// Per-step host<->device transfers go through pinned (GPU-addressable) staging and this kernel instead of
// hipMemcpyAsync: the copy-engine path stalled the stream for tens of ms every dozen steps on the test box.
__global__ void copy_words_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, int64_t nwords) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// bulk variant for 16-byte aligned blocks (parameter staging -> device): one dwordx4 per lane, a wave moves 1 KiB per request
__global__ void copy_vec4_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, int64_t nvec) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
hipError_t launch_copy_bytes16(void *dst, const void *src, size_t bytes, hipStream_t s) {
    const int64_t nv = (int64_t)((bytes + 15) / 16);
    if (nv == 0) return hipSuccess;
    DPMM_LAUNCH(copy_vec4_kernel, dim3(grid_for(nv)), dim3(256), 0, s, (uint4 *)dst, (const uint4 *)src, nv);
    return hipGetLastError();
}
hipError_t launch_copy_bytes(void *dst, const void *src, size_t bytes, hipStream_t s) {
    const int64_t nw = (int64_t)((bytes + 3) / 4);
    if (nw == 0) return hipSuccess;
    DPMM_LAUNCH(copy_words_kernel, dim3(grid_for(nw)), dim3(256), 0, s, (uint32_t *)dst, (const uint32_t *)src, nw);
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
    DPMM_LAUNCH(contingency_kernel, dim3(grid_for(n)), dim3(256), lds, s, bins, gt, n, K, n_gt, counts);
    return hipGetLastError();
}
__global__ void i64_to_i32_kernel(int32_t *dst, const int64_t *src, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (int32_t)src[i];
}
hipError_t launch_i64_to_i32(int32_t *dst, const int64_t *src, int64_t n, hipStream_t s) {
    DPMM_LAUNCH(i64_to_i32_kernel, dim3(grid_for(n)), dim3(256), 0, s, dst, src, n);
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
        DPMM_LAUNCH((ingest_rows_kernel<double>), dim3(grid_for(rows * D)), dim3(256), 0, s, dst, ldx, (const double *)src, ld, rows, D, nan_to_zero);
    else
        DPMM_LAUNCH((ingest_rows_kernel<float>), dim3(grid_for(rows * D)), dim3(256), 0, s, dst, ldx, (const float *)src, ld, rows, D, nan_to_zero);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Smart splits (src/local_clusters_actions.jl:555-653): 1-D 2-means along a direction v inside one cluster.
//   tranform_points_worker! (:643-653)   t_i = v . (x_i - mu), Float64, for the points with label == k
//   kmeans_iter_worker!     (:635-641)   side_i = |t_i - m_lo| < |t_i - m_hi| ? 1 : 2; per side (sum of t, count)
//   set_smart_labels_in_worker! (:629-633) sub-label_i = side_i
// proj[i] is indexed by point (only entries of the cluster are meaningful); `vals` receives the same values compacted
// in arbitrary order (the host takes percentiles of them).  The per-side sums are reduced in a FIXED order (one partial
// per workgroup over a contiguous range, tree inside the workgroup): reproducible run to run.
__global__ __launch_bounds__(256) void smart_project_kernel(const int32_t *__restrict__ bins, const float *__restrict__ X, int64_t ldx, int64_t n,
                                                            int D, int k, const double *__restrict__ v, const double *__restrict__ mu,
                                                            double *__restrict__ proj, double *__restrict__ vals, unsigned long long *counter) {
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        const bool mine = i < n && (bins[i] >> 1) == k;
        double t = 0.;
        if (mine) {
            const float *x = X + i * ldx;
            for (int d = 0; d < D; ++d) t += v[d] * ((double)x[d] - mu[d]);       // v' * (x - mu), left to right as a dot product
            proj[i] = t;
        }
        const unsigned long long m = __ballot(mine);
        if (m) {
            const int lane = threadIdx.x & 63;
            unsigned long long base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, (unsigned long long)__popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (mine) vals[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
        }
    }
}

constexpr int SMART_GROUPS = 1024;
__global__ __launch_bounds__(256) void smart_kmeans_kernel(const int32_t *__restrict__ bins, const double *__restrict__ proj, int64_t n, int k,
                                                           double m_lo, double m_hi, double *__restrict__ partial) {
    __shared__ double sh[4][256];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(n, lo + per);
    double s1 = 0., c1 = 0., s2 = 0., c2 = 0.;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        if ((bins[i] >> 1) == k) {
            const double t = proj[i];
            if (fabs(t - m_lo) < fabs(t - m_hi)) { s1 += t; c1 += 1.; } else { s2 += t; c2 += 1.; }
        }
    }
    sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = c1; sh[2][threadIdx.x] = s2; sh[3][threadIdx.x] = c2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[4 * blockIdx.x + threadIdx.x] = sh[threadIdx.x][0];
}
__global__ void smart_kmeans_finish_kernel(const double *__restrict__ partial, int groups, double *__restrict__ out) {
    if (threadIdx.x < 4) {
        double s = 0.;
        for (int g = 0; g < groups; ++g) s += partial[4 * g + threadIdx.x];
        out[threadIdx.x] = s;
    }
}
__global__ void smart_assign_kernel(int32_t *__restrict__ bins, const double *__restrict__ proj, int64_t n, int k, double m_lo, double m_hi) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = bins[i];
        if ((b >> 1) == k) {
            const double t = proj[i];
            bins[i] = 2 * k + ((fabs(t - m_lo) < fabs(t - m_hi)) ? 0 : 1);
        }
    }
}
hipError_t launch_smart_project(const int32_t *bins, const float *X, int64_t ldx, int64_t n, int D, int k, const double *v, const double *mu,
                                double *proj, double *vals, unsigned long long *counter, hipStream_t s) {
    DPMM_LAUNCH(smart_project_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, X, ldx, n, D, k, v, mu, proj, vals, counter);
    return hipGetLastError();
}
hipError_t launch_smart_kmeans(const int32_t *bins, const double *proj, int64_t n, int k, double m_lo, double m_hi, double *partial, double *out,
                               hipStream_t s) {
    DPMM_LAUNCH(smart_kmeans_kernel, dim3(SMART_GROUPS), dim3(256), 0, s, bins, proj, n, k, m_lo, m_hi, partial);
    DPMM_LAUNCH(smart_kmeans_finish_kernel, dim3(1), dim3(64), 0, s, partial, SMART_GROUPS, out);
    return hipGetLastError();
}
hipError_t launch_smart_assign(int32_t *bins, const double *proj, int64_t n, int k, double m_lo, double m_hi, hipStream_t s) {
    DPMM_LAUNCH(smart_assign_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, proj, n, k, m_lo, m_hi);
    return hipGetLastError();
}
int smart_groups() { return SMART_GROUPS; }

// predict_points (src/local_clusters_actions.jl:23-40) finish on the device: row-wise argmax (Julia semantics: the first NaN wins,
// else the first maximum), NaN -> -inf, subtract the row maximum, exp, divide by the row sum.  table[row(k)][i], row(k) = k * rstep.
__global__ void predict_finish_kernel(const float *__restrict__ table, int64_t stride, int rstep, int64_t n, int K, int64_t *__restrict__ labels,
                                      float *__restrict__ probs) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float m = -INFINITY;
        int best = 0;
        bool nan_seen = false;
        for (int k = 0; k < K; ++k) {
            const float a = table[(int64_t)k * rstep * stride + i];
            if (a != a) {
                if (!nan_seen) { nan_seen = true; best = k; }
            } else if (a > m) {
                m = a;
                if (!nan_seen) best = k;
            }
        }
        labels[i] = best + 1;
        if (probs) {
            float s = 0.f;
            float *pr = probs + i * K;
            for (int k = 0; k < K; ++k) {
                float a = table[(int64_t)k * rstep * stride + i];
                if (a != a) a = -INFINITY;
                const float e = expf(a - m);
                pr[k] = e;
                s += e;
            }
            for (int k = 0; k < K; ++k) pr[k] = pr[k] / s;
        }
    }
}
hipError_t launch_predict_finish(const float *table, int64_t stride, int rstep, int64_t n, int K, int64_t *labels, float *probs, hipStream_t s) {
    DPMM_LAUNCH(predict_finish_kernel, dim3(grid_for(n)), dim3(256), 0, s, table, stride, rstep, n, K, labels, probs);
    return hipGetLastError();
}

hipError_t launch_init_labels(int32_t *bins, int64_t n, int64_t first, int init_clusters, int label0, uint64_t seed, uint32_t epoch, hipStream_t s) {
    DPMM_LAUNCH(init_labels_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, first, init_clusters, label0, seed, epoch);
    return hipGetLastError();
}
hipError_t launch_bins_from_i64(int32_t *bins, const int64_t *labels, const int64_t *sub, int64_t n, hipStream_t s) {
    DPMM_LAUNCH(bins_from_i64_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, labels, sub, n);
    return hipGetLastError();
}
hipError_t launch_bins_to_i64(const int32_t *bins, int64_t *labels, int64_t *sub, int64_t n, hipStream_t s) {
    DPMM_LAUNCH(bins_to_i64_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, labels, sub, n);
    return hipGetLastError();
}
hipError_t launch_split(int32_t *bins, int64_t n, int64_t first, const int32_t *pairs, int m, uint64_t seed, uint32_t epoch, hipStream_t s) {
    DPMM_LAUNCH(split_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, first, pairs, m, seed, epoch);
    return hipGetLastError();
}
hipError_t launch_merge(int32_t *bins, int64_t n, const int32_t *pairs, int m, hipStream_t s) {
    DPMM_LAUNCH(merge_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, pairs, m);
    return hipGetLastError();
}
hipError_t launch_remap(int32_t *bins, int64_t n, const int32_t *map, hipStream_t s) {
    DPMM_LAUNCH(remap_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, map);
    return hipGetLastError();
}
hipError_t launch_reset_sub(int32_t *bins, int64_t n, int64_t first, const int32_t *idx, int m, uint64_t seed, uint32_t epoch, hipStream_t s) {
    DPMM_LAUNCH(reset_sub_kernel, dim3(grid_for(n)), dim3(256), 0, s, bins, n, first, idx, m, seed, epoch);
    return hipGetLastError();
}

}  // namespace dpmm
