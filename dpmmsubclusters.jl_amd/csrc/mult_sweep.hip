// mult_sweep.hip -- fused label + sub-label sampling for the Multinomial prior on gfx950.
//
// Stands in for (reference paths relative to the reference checkout):
//   sample_labels_worker!               src/local_clusters_actions.jl:112-134
//   log_likelihood!(::multinomial_dist) src/distributions/multinomial_dist.jl:13-15  (r_i = alpha' x_i)
//   sample_log_cat_array!               src/utils.jl:19-31
//   sample_sub_clusters_worker! / create_subclusters_labels!   src/local_clusters_actions.jl:70-95
//
// Regime: D ~ 1000 dense Float32 counts, 4*D bytes per point against 2*D*(K+2) flops: HBM
// streaming with a skinny GEMM on top.  The x stream is read once; all 3K log-probability
// rows (cluster, left, right for every cluster) are contracted against it on the FP32 matrix
// cores (v_mfma_f32_16x16x4_f32, M = 16 parameter rows, N = 16 points, K = 4 features), so the
// sub-label phase needs no second pass over x and no dependence on the label mix of a tile.
//   A operand: logp rows, pre-packed fragment image Lp[rowblock][t][lane][4] (L2 resident)
//   B operand: x, float4 per lane straight from HBM: lane (c, g) reads elements 16t+4g.. of point c
// The 3K x TILE table (+ log weights) goes to a per-workgroup scratch (L2), then every lane
// draws label and sub-label for one point with the oracle-identical inverse-CDF scan.
#include <type_traits>
#include "dpmm_device.h"
#include "dpmm_kernels.h"

namespace dpmm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int M_NG = 4;        // 16-point groups per wave
constexpr int M_TILE = 256;    // points per workgroup
constexpr int M_RBP = 8;       // row blocks (of 16 parameter rows) per pass over the features

__global__ __launch_bounds__(256) void mult_sweep_kernel(MultSweepArgs A, const float *__restrict__ Lp, int NT, int NRB) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane & 15, g = lane >> 4;
    const int K = A.K, rows = 3 * K;
    const int64_t ntiles = (A.n + M_TILE - 1) / M_TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = tile * M_TILE + (int64_t)wave * 64;
        float *scr = A.scratch + (A.scratch_by_tile ? tile * M_TILE : (int64_t)blockIdx.x * M_TILE) + wave * 64;
        const int64_t sstride = A.scratch_stride;
        const float *xp[M_NG];
        bool pv[M_NG];
#pragma unroll
        for (int n = 0; n < M_NG; ++n) {
            const int64_t p = wbase + 16 * n + ci;
            pv[n] = p < A.n;
            xp[n] = A.X + (pv[n] ? p : 0) * A.ldx + 4 * g;
        }
        for (int rb0 = 0; rb0 < NRB; rb0 += M_RBP) {
            f32x4 acc[M_RBP][M_NG];
#pragma unroll
            for (int rb = 0; rb < M_RBP; ++rb)
#pragma unroll
                for (int n = 0; n < M_NG; ++n) acc[rb][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int nrb = min(M_RBP, NRB - rb0);
#pragma unroll 2
            for (int t = 0; t < NT; ++t) {
                f32x4 x[M_NG];
                const bool ev = 16 * t + 4 * g < A.ldx;
#pragma unroll
                for (int n = 0; n < M_NG; ++n)
                    x[n] = (pv[n] && ev) ? *reinterpret_cast<const f32x4 *>(xp[n] + 16 * t) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rb = 0; rb < M_RBP; ++rb) {
                    if (rb < nrb) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(Lp + ((size_t)(rb0 + rb) * NT + t) * 256 + lane * 4);
#pragma unroll
                        for (int n = 0; n < M_NG; ++n) {
                            acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x[n].x, acc[rb][n], 0, 0, 0);
                            acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x[n].y, acc[rb][n], 0, 0, 0);
                            acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x[n].z, acc[rb][n], 0, 0, 0);
                            acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x[n].w, acc[rb][n], 0, 0, 0);
                        }
                    }
                }
            }
            // C layout: reg r -> parameter row 16(rb0+rb) + 4g + r, point 16n + ci
#pragma unroll
            for (int rb = 0; rb < M_RBP; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * (rb0 + rb) + 4 * g + r;
                    if (rb < nrb && row < rows) {
                        const float cst = A.cst[row];
#pragma unroll
                        for (int n = 0; n < M_NG; ++n) scr[(int64_t)row * sstride + 16 * n + ci] = acc[rb][n][r] + cst;
                    }
                }
        }
        __syncthreads();  // table rows were written by other lanes of the wave (workgroup-scope fence)
        const int64_t myp = wbase + lane;
        const bool valid = myp < A.n;
        if (valid && !A.labels_only) {
            const float *col = scr + lane;
            const Philox4 rr = philox4x32_10(A.seed, (uint64_t)(A.first_index + myp), A.epoch, STREAM_SWEEP);
            int z = 0;
            float m = -INFINITY;
            int best = 0;
            bool nan_seen = false;
            for (int k = 0; k < K; ++k) {
                const float a = col[(int64_t)(3 * k) * sstride];
                if (a != a) {
                    if (!nan_seen) { nan_seen = true; best = k; }
                } else if (a > m) {
                    m = a;
                    if (!nan_seen) best = k;
                }
            }
            if (A.final_argmax) {
                z = best;
            } else if (m == -INFINITY) {
                z = 0;
            } else {
                float s = 0.f;
                for (int k = 0; k < K; ++k) s += exp_det(nan_to_ninf(col[(int64_t)(3 * k) * sstride]) - m);
                const float t = u01(rr.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                for (int k = 0; k < K; ++k) {
                    cw += exp_det(nan_to_ninf(col[(int64_t)(3 * k) * sstride]) - m);
                    if (!(cw < t)) { z = k; break; }
                }
            }
            const float b0 = col[(int64_t)(3 * z + 1) * sstride], b1 = col[(int64_t)(3 * z + 2) * sstride];
            A.bins[myp] = 2 * z + draw2(b0, b1, u01(rr.v[1]));
        }
        __syncthreads();  // scratch rows are reused by the next tile
    }
}

// Lp[rb][t][lane][jj] = logp[16 rb + (lane & 15)][16 t + 4 (lane >> 4) + jj]  (zero outside [3K) x [ldx))
__global__ void mult_pack_kernel(const float *__restrict__ logp, float *__restrict__ Lp, int rows, int64_t ldx, int NT, int NRB) {
    const int64_t total = (int64_t)NRB * NT * 256;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int jj = (int)(e & 3), lane = (int)((e >> 2) & 63);
        const int64_t bt = e >> 8;
        const int t = (int)(bt % NT), rb = (int)(bt / NT);
        const int row = 16 * rb + (lane & 15), col = 16 * t + 4 * (lane >> 4) + jj;
        Lp[e] = (row < rows && col < ldx) ? logp[(size_t)row * ldx + col] : 0.f;
    }
}


// ---------------------------------------------------------------------------------------
// bf16 path.  Multinomial observations are counts: whenever every x is exactly representable in
// bf16 (checked once at upload; integers up to 256 are) the products are formed on the bf16 matrix
// cores, 16x the FP32-MFMA rate.  Each Float32 log-probability is split exactly into three bf16
// terms (hi + mid + lo, 8 significant bits each), every bf16 x bf16 product is exact in the f32
// accumulator, so the table differs from the f32-MFMA one only by summation order.
//   v_mfma_f32_16x16x32_bf16: lane (i = l & 15, g = l >> 4): A[row i][k = 8g + j], B[k = 8g + j][col i], j = 0..7;
//   C/D reg r: row 4g + r, col i.
// A chunk for one k-step (32 features) = NRB x 3 planes x 1 KiB, staged global -> registers -> LDS one
// k-step ahead and shared by the 4 waves; x (f32 in HBM) is loaded one k-step ahead and truncated to bf16.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ u32x4 pack_bf16x8(const f32x4 lo, const f32x4 hi) {
    u32x4 r;
    r.x = (__float_as_uint(lo.x) >> 16) | (__float_as_uint(lo.y) & 0xffff0000u);
    r.y = (__float_as_uint(lo.z) >> 16) | (__float_as_uint(lo.w) & 0xffff0000u);
    r.z = (__float_as_uint(hi.x) >> 16) | (__float_as_uint(hi.y) & 0xffff0000u);
    r.w = (__float_as_uint(hi.z) >> 16) | (__float_as_uint(hi.w) & 0xffff0000u);
    return r;
}

// ---------------------------------------------------------------------------------------
// u8 path.  Bag-of-words data are small non-negative integers: when every x is an integer in [0, 255] (checked at upload) the
// context also keeps the points as BYTES ([n][ld8], ld8 = roundup(D, 128)) -- a lossless re-encoding of the Float32 matrix the
// caller handed over -- and the sweep streams 1 byte per element instead of 4.  A k-step covers 128 features (= one 128-byte
// line per point): lane (i, g) loads the 32 bytes [32 g, 32 g + 32) of point i's line with two 16-byte loads, and the four
// v_mfma_f32_16x16x32_bf16 "slices" of the k-step contract features {128 ks + 32 g + 8 sl + j}; the parameter planes are packed
// in exactly that order (mult_pack_u8_kernel), so each lane's bytes are contiguous.  byte -> bf16 is exact (8 significant bits).
// The slice loop re-uses the bf16 kernel's LDS double buffer (one 1-KiB fragment per row block and plane); x is fetched ONE
// k-step = four slices ahead.  Same table, same draw code, bit-identical labels (only the summation order over features differs
// from the Float32-loading kernel, as it already did between the two older kernels).
__device__ __forceinline__ u32x4 bytes_to_bf16x8(uint32_t w0, uint32_t w1) {
    auto two = [](uint32_t lo_byte, uint32_t hi_byte) -> uint32_t {
        const float a = (float)lo_byte, b = (float)hi_byte;          // v_cvt_f32_ubyteN
        return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
    };
    u32x4 r;
    r.x = two(w0 & 0xffu, (w0 >> 8) & 0xffu);
    r.y = two((w0 >> 16) & 0xffu, w0 >> 24);
    r.z = two(w1 & 0xffu, (w1 >> 8) & 0xffu);
    r.w = two((w1 >> 16) & 0xffu, w1 >> 24);
    return r;
}

// Which parameter rows a tile needs (round 3).  A point's draw reads the K cluster-level values and, of the 2K sub-cluster values, only
// the TWO of the cluster it lands in -- and in a running chain that is almost always the cluster it was in.  The u8 planes are therefore
// packed in two groups of 16-row blocks (mult_pack_u8_kernel): blocks [0, NRBc) hold the K cluster rows, blocks NRBc + j the (left, right)
// rows of clusters 8j .. 8j+7.  A tile evaluates the cluster blocks plus the sub-cluster blocks of the clusters its points were in (a
// 128-bit set built from the previous labels; with the bin-sorted visiting order a tile spans one or two of them), draws the labels, and
// runs a second pass over its points only for sub-cluster blocks a NEW label asks for that the first pass did not cover (rare).  A row's
// value does not depend on which other rows share its pass (each 16-row block accumulates over the features in the same order), so the
// labels are the ones the all-rows kernel drew; table mode (debug tables, predict) evaluates every block.  K = 32: 3-4 blocks per tile
// instead of 6.
template <int B_RBP>
__global__ __launch_bounds__(256, (B_RBP <= 4 ? 2 : 1)) void mult_sweep_u8_kernel(MultSweepArgs A, const uint8_t *__restrict__ X8, int64_t ld8,
                                                                                   const uint32_t *__restrict__ Lp8, int NKS8, int NRBc, int NRBs,
                                                                                   int ltab_ok) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    // two fragment buffers (slice sl in buffer sl % 2); between the passes of a tile the same memory holds the tile's K x 256 cluster-level
    // values (when they fit and the tile needs one pass)
    constexpr int BUFW = B_RBP * 3 * 256;
    float *const ltab_all = reinterpret_cast<float *>(dyn_lds);
    __shared__ uint32_t need[4], miss[4];
    __shared__ int blist[64 + 128];
    __shared__ int nlist;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane & 15, g = lane >> 4;
    const int K = A.K, NRB = NRBc + NRBs;
    const int64_t ntiles = (A.n + M_TILE - 1) / M_TILE;
    const bool use_order = A.order != nullptr && !A.labels_only && *A.order_total == (int32_t)A.n;
    const bool tab_fits = ltab_ok && !A.labels_only;
    float *const ltab = ltab_all + wave * 64;                  // column of point `lane` of this wave: ltab[k * 256 + lane]
#ifdef DPMM_U8_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
#define U8_STAMP(i) { const unsigned long long t_now = __builtin_readcyclecounter(); st_acc[i] += t_now - t_last; t_last = t_now; }
#else
#define U8_STAMP(i)
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef DPMM_U8_STAMPS
        unsigned long long t_last = __builtin_readcyclecounter();
#endif
        const int64_t i0 = tile * M_TILE + (int64_t)wave * 64;
        float *scr = A.scratch + (A.scratch_by_tile ? tile * M_TILE : (int64_t)blockIdx.x * M_TILE) + wave * 64;
        const int64_t sstride = A.scratch_stride;
        const bool valid = i0 + lane < A.n;
        const int myp32 = valid ? (use_order ? A.order[i0 + lane] : (int)(i0 + lane)) : 0;     // (table mode keeps storage order: column = point)
        const int64_t myp = A.labels_only ? i0 + lane : (int64_t)myp32;
        const uint8_t *xp0[M_NG];
        bool pv[M_NG];
#pragma unroll
        for (int n = 0; n < M_NG; ++n) {
            const int64_t p = A.labels_only ? i0 + 16 * n + ci : (int64_t)__shfl(myp32, 16 * n + ci);
            pv[n] = i0 + 16 * n + ci < A.n;
            xp0[n] = X8 + (pv[n] ? p : 0) * ld8 + 32 * g;     // row of a valid point (point 0 for the padding lanes), this lane's 32 bytes
        }
        bool tab_lds = false;                                  // this tile's cluster-level values are in LDS
        if (tid < 4) { need[tid] = 0u; miss[tid] = 0u; }
        __syncthreads();
        if (valid && !A.labels_only && A.use_prev) {
            const int zp = A.bins[myp] >> 1;
            if (zp >= 0 && zp < K) atomicOr(&need[zp >> 8], 1u << ((zp >> 3) & 31));
        }
        __syncthreads();
        if (tid == 0) {
            int c = 0;
            for (int b = 0; b < NRBc; ++b) blist[c++] = b;
            for (int j = 0; j < NRBs; ++j)
                if (A.labels_only || ((need[j >> 5] >> (j & 31)) & 1u)) blist[c++] = NRBc + j;
            nlist = c;
        }
        __syncthreads();
        // one or more passes over the features for the row blocks blist[0 .. cnt)
        auto run_passes = [&](const int cnt, const bool first_call) {
            for (int rb0 = 0; rb0 < cnt; rb0 += B_RBP) {
                const int nrb = min(B_RBP, cnt - rb0);
                f32x4 acc[B_RBP][M_NG];
#pragma unroll
                for (int rb = 0; rb < B_RBP; ++rb)
#pragma unroll
                    for (int n = 0; n < M_NG; ++n) acc[rb][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // The pass body is compiled once per number of row blocks NR (1 .. B_RBP): with NR a constant the fragment reads of a slice
                // are a straight line that runs two reads ahead of the matrix instructions.  (With `if (rb < nrb)` around each block the
                // compiler emitted read -> wait -> 4 MFMA per plane: 9-12 exposed LDS latencies per slice, 2.3 k of a slice's 2.7 k cycles.)
                auto pass = [&](auto NRc) {
                    constexpr int NR = decltype(NRc)::value;
                    constexpr int CV4 = NR * 192;                  // 16-byte vectors of a slice's fragments: 192 per row block (3 planes x 64 lanes)
                    constexpr int NST = (CV4 + 255) / 256;
                    const int NSL = 4 * NKS8;                     // slices = LDS chunks of this pass
                    u32x4 st[NST];
                    int boff[NST];                                // where staged vector p of this thread comes from
#pragma unroll
                    for (int p = 0; p < NST; ++p) {
                        const int e = min(p * 256 + tid, CV4 - 1);
                        boff[p] = blist[rb0 + e / 192] * 192 + e % 192;
                    }
                    auto prefetch = [&](int sl) {                 // block b of slice sl lives at Lp8 + (sl * NRB + b) * 768 words
                        const u32x4 *src = reinterpret_cast<const u32x4 *>(Lp8) + (size_t)sl * NRB * 192;
#pragma unroll
                        for (int p = 0; p < NST; ++p) st[p] = src[boff[p]];
                    };
                    auto commit = [&](uint32_t *buf) {
#pragma unroll
                        for (int p = 0; p < NST; ++p) reinterpret_cast<u32x4 *>(buf)[min(p * 256 + tid, CV4 - 1)] = st[p];
                    };
                    auto loadx = [&](int ks, u32x4 (&xa)[M_NG], u32x4 (&xb)[M_NG]) {     // unconditional: ld8 is a multiple of 128, padding is zero
#pragma unroll
                        for (int n = 0; n < M_NG; ++n) {
                            xa[n] = *reinterpret_cast<const u32x4 *>(xp0[n] + 128 * (int64_t)ks);
                            xb[n] = *reinterpret_cast<const u32x4 *>(xp0[n] + 128 * (int64_t)ks + 16);
                        }
                    };
                    u32x4 xa[M_NG], xb[M_NG], na[M_NG], nb[M_NG];
                    // Slice sl is staged one slice AHEAD of its use: during slice sl every thread commits its part of slice sl + 1 to the other
                    // buffer, and the workgroup barrier sits at the END of the slice.  A wave that leaves the barrier finds the fragments of its
                    // next slice complete in LDS -- the chain global load -> ds_write -> barrier -> ds_read -> MFMA of the earlier scheme (the
                    // barrier between the write and the read of the SAME slice) is off the critical path; the barrier only absorbs the skew of
                    // the waves' matrix phases.  Two buffers still do: the buffer slice sl + 1 goes to was last read in slice sl - 1, and every
                    // wave finished that before the barrier this wave has already passed.
                    prefetch(0);
                    loadx(0, xa, xb);
                    int cur = 0;
                    commit(dyn_lds);
                    if (NSL > 1) prefetch(1);
                    __syncthreads();
                    for (int ks = 0; ks < NKS8; ++ks) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int sl = 4 * ks + q;
                            uint32_t *buf = dyn_lds + cur * BUFW;
                            cur ^= 1;
                            u32x4 a[3];
                            auto rd = [&](int i) { return *reinterpret_cast<const u32x4 *>(buf + i * 256 + lane * 4); };   // i = 3 rb + plane
                            a[0] = rd(0);
                            a[1] = rd(1);
                            if (sl + 1 < NSL) commit(dyn_lds + cur * BUFW);     // slice sl + 1 (in registers since the previous slice) -> next buffer
                            if (sl + 2 < NSL) prefetch(sl + 2);
                            // next k-step's bytes.  Vector loads return in order: issued BEHIND this slice's fragment request they stay in
                            // flight across the next commit (which waits for that request only) -- two slices of cover instead of one
                            if (q == 0) loadx(min(ks + 1, NKS8 - 1), na, nb);
                            u32x4 xq[M_NG];
#pragma unroll
                            for (int n = 0; n < M_NG; ++n) {
                                const u32x4 src = q < 2 ? xa[n] : xb[n];
                                const uint32_t w0 = (q & 1) ? src.z : src.x, w1 = (q & 1) ? src.w : src.y;
                                xq[n] = pv[n] ? bytes_to_bf16x8(w0, w1) : (u32x4){0u, 0u, 0u, 0u};
                            }
#pragma unroll
                            for (int i = 0; i < 3 * NR; ++i) {
                                if (i + 2 < 3 * NR) a[(i + 2) % 3] = rd(i + 2);
#pragma unroll
                                for (int n = 0; n < M_NG; ++n)
                                    acc[i / 3][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i % 3]), __builtin_bit_cast(bf16x8, xq[n]), acc[i / 3][n], 0, 0, 0);
                            }
                            __syncthreads();
                        }
#pragma unroll
                        for (int n = 0; n < M_NG; ++n) { xa[n] = na[n]; xb[n] = nb[n]; }
                    }
                };
                switch (nrb) {
                    case 1: pass(std::integral_constant<int, 1>{}); break;
                    case 2: pass(std::integral_constant<int, 2>{}); break;
                    case 3: if constexpr (B_RBP >= 3) pass(std::integral_constant<int, 3>{}); break;
                    case 4: if constexpr (B_RBP >= 4) pass(std::integral_constant<int, 4>{}); break;
                    case 5: if constexpr (B_RBP >= 5) pass(std::integral_constant<int, 5>{}); break;
                    case 6: if constexpr (B_RBP >= 6) pass(std::integral_constant<int, 6>{}); break;
                    case 7: if constexpr (B_RBP >= 7) pass(std::integral_constant<int, 7>{}); break;
                    default: if constexpr (B_RBP >= 8) pass(std::integral_constant<int, 8>{}); break;
                }
                // (the last slice ended with a barrier: every wave is done with the fragment buffers of this pass)
                const bool tab_now = tab_fits && first_call && cnt <= B_RBP;
                if (first_call) tab_lds = tab_now;
#pragma unroll
                for (int rb = 0; rb < B_RBP; ++rb) {
                    if (rb < nrb) {
                        const int vb = blist[rb0 + rb];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 4 * g + r;
                            const int k = NRBs == 0 ? i / 3 : (vb < NRBc ? 16 * vb + i : 8 * (vb - NRBc) + (i >> 1));
                            const int w = NRBs == 0 ? i % 3 : (vb < NRBc ? 0 : 1 + (i & 1));
                            if (k < K) {
                                const int row = 3 * k + w;
                                const float cst = A.cst[row];
#pragma unroll
                                for (int n = 0; n < M_NG; ++n) {
                                    const float v = acc[rb][n][r] + cst;
                                    if (w == 0 && tab_now) ltab[k * 256 + 16 * n + ci] = v;      // cluster-level rows: the label draw reads them from LDS
                                    else scr[(int64_t)row * sstride + 16 * n + ci] = v;
                                }
                            }
                        }
                    }
                }
            }
        };
        U8_STAMP(0)
        run_passes(nlist, true);
        __syncthreads();
        U8_STAMP(1)
        int z = 0;
        const Philox4 rr = philox4x32_10(A.seed, (uint64_t)(A.first_index + myp), A.epoch, STREAM_SWEEP);
        if (valid && !A.labels_only) {
            // same arithmetic, same order as the other Multinomial kernels (and the CPU oracle); the K values of a point come from LDS when
            // they fit (the global scratch cost ~30 % of the kernel: three passes of K dependent-latency L2 loads per point)
            const float *col = tab_lds ? ltab + lane : scr + lane;
            const int64_t cs = tab_lds ? 256 : 3 * sstride;
            float e[32];
            bool in_regs = false;
            if (tab_lds && K <= 32 && !A.final_argmax) {
                // the point's K values in registers: one batch of LDS reads, each exponential taken once, no data-dependent exit (the three
                // dependent-latency passes below were 16 % of the kernel at K = 32).  Rows past K count as -inf; sums run in
                // the same order over the same values (+0.f past K), and the running sum is monotone, so the first k with !(cw < t) is the
                // number of k with cw < t.  A NaN anywhere (never in practice) leaves the point to the general loops.
                bool nan_any = false;
                float m = -INFINITY;
                int Kv = K;
                asm volatile("" : "+v"(Kv));                       // per-lane compare: 32 uniform conditions would be hoisted out of the tile loop into SGPRs
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const float a = col[k * 256];                    // rows past K: whatever the fragment buffers left there (the launcher allocates 32 rows)
                    e[k] = k < Kv ? a : -INFINITY;
                    nan_any |= e[k] != e[k];
                    m = fmaxf(m, e[k]);
                }
                if (!nan_any) {
                    in_regs = true;
                    if (m == -INFINITY) {
                        z = 0;
                    } else {
                        float s = 0.f;
#pragma unroll
                        for (int k = 0; k < 32; ++k) {
                            e[k] = exp_det(e[k] - m);
                            s += e[k];
                        }
                        const float t = u01(rr.v[0]) * s;
                        float cw = 0.f;
                        int below = 0;
#pragma unroll
                        for (int k = 0; k < 32; ++k) {
                            cw += e[k];
                            below += cw < t ? 1 : 0;
                        }
                        z = below < K - 1 ? below : K - 1;
                    }
                }
            }
            if (!in_regs) {
            float m = -INFINITY;
            int best = 0;
            bool nan_seen = false;
            for (int k = 0; k < K; ++k) {
                const float a = col[k * cs];
                if (a != a) {
                    if (!nan_seen) { nan_seen = true; best = k; }
                } else if (a > m) {
                    m = a;
                    if (!nan_seen) best = k;
                }
            }
            if (A.final_argmax) {
                z = best;
            } else if (m == -INFINITY) {
                z = 0;
            } else {
                float s = 0.f;
                for (int k = 0; k < K; ++k) s += exp_det(nan_to_ninf(col[k * cs]) - m);
                const float t = u01(rr.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                for (int k = 0; k < K; ++k) {
                    cw += exp_det(nan_to_ninf(col[k * cs]) - m);
                    if (!(cw < t)) { z = k; break; }
                }
            }
            }
            const int j = z >> 3;
            if (NRBs > 0 && !((need[j >> 5] >> (j & 31)) & 1u)) atomicOr(&miss[j >> 5], 1u << (j & 31));
        }
        if (!A.labels_only) {
            __syncthreads();
            U8_STAMP(2)
            if (miss[0] | miss[1] | miss[2] | miss[3]) {           // (uniform) a new label outside the blocks of the first pass: evaluate those now
                __syncthreads();
                if (tid == 0) {
                    int c = 0;
                    for (int j2 = 0; j2 < NRBs; ++j2)
                        if ((miss[j2 >> 5] >> (j2 & 31)) & 1u) blist[c++] = NRBc + j2;
                    nlist = c;
                }
                __syncthreads();
                run_passes(nlist, false);
                __syncthreads();
            }
            if (valid) {
                const float *gcol = scr + lane;
                const float b0 = gcol[(int64_t)(3 * z + 1) * sstride], b1 = gcol[(int64_t)(3 * z + 2) * sstride];
                A.bins[myp] = 2 * z + draw2(b0, b1, u01(rr.v[1]));
            }
        }
        __syncthreads();
        U8_STAMP(3)
    }
#ifdef DPMM_U8_STAMPS
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 77 || blockIdx.x == 300) && !A.labels_only)
        printf("u8 stamps block %d: setup %llu passes %llu draw %llu subdraw+fallback %llu\n", (int)blockIdx.x, st_acc[0], st_acc[1], st_acc[2], st_acc[3]);
#endif
}

template <int B_RBP>   // row blocks (16 parameter rows each) per pass over the features
__global__ __launch_bounds__(256, (B_RBP <= 6 ? 2 : 1)) void mult_sweep_bf16_kernel(MultSweepArgs A, const uint32_t *__restrict__ Lp16, int NKS, int NRB) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][B_RBP * 3 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane & 15, g = lane >> 4;
    const int K = A.K, rows = 3 * K;
    const int64_t ntiles = (A.n + M_TILE - 1) / M_TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = tile * M_TILE + (int64_t)wave * 64;
        float *scr = A.scratch + (A.scratch_by_tile ? tile * M_TILE : (int64_t)blockIdx.x * M_TILE) + wave * 64;
        const int64_t sstride = A.scratch_stride;
        const float *xp0[M_NG];
        bool pv[M_NG];
#pragma unroll
        for (int n = 0; n < M_NG; ++n) {
            const int64_t p = wbase + 16 * n + ci;
            pv[n] = p < A.n;
            xp0[n] = A.X + (pv[n] ? p : 0) * A.ldx;          // row of a valid point (point 0 for the padding lanes)
        }
        for (int rb0 = 0; rb0 < NRB; rb0 += B_RBP) {
            const int nrb = min(B_RBP, NRB - rb0);
            const int chunk_words = nrb * 3 * 256;                // uint32 words per k-step chunk
            f32x4 acc[B_RBP][M_NG];
#pragma unroll
            for (int rb = 0; rb < B_RBP; ++rb)
#pragma unroll
                for (int n = 0; n < M_NG; ++n) acc[rb][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // chunk(ks) lives at Lp16 + (ks * NRB + rb0) * 768 words; the nrb row blocks of a pass are contiguous.
            // All loads are UNCONDITIONAL (clamped addresses / indices; masking happens when the values are consumed one
            // k-step later): a conditional load compiles to a branch with `s_waitcnt vmcnt(0)` behind it.
            // (Measured and dropped: 128-point tiles at three workgroups per CU, a two-k-step x prefetch (both spill), and one
            // workgroup per CU with x three k-steps ahead in registers: 1.70 ms against 1.36 ms.)
            constexpr int NST = (B_RBP * 3 * 256 / 4 + 255) / 256;
            const int chunk_v4 = chunk_words / 4;
            u32x4 st[NST];
            auto prefetch = [&](int ks) {
                const u32x4 *src = reinterpret_cast<const u32x4 *>(Lp16 + ((size_t)ks * NRB + rb0) * 768);
#pragma unroll
                for (int p = 0; p < NST; ++p) st[p] = src[min(p * 256 + tid, chunk_v4 - 1)];     // surplus threads re-read the last vector
            };
            auto commit = [&](uint32_t *buf) {
#pragma unroll
                for (int p = 0; p < NST; ++p) reinterpret_cast<u32x4 *>(buf)[min(p * 256 + tid, chunk_v4 - 1)] = st[p];   // ... and re-write it
            };
            auto loadx = [&](int ks, f32x4 (&xl)[M_NG], f32x4 (&xh)[M_NG]) {
                const int e = 32 * ks + 8 * g;
                const int el = e < A.ldx ? e : 0, eh = e + 4 < A.ldx ? e + 4 : 0;      // in-row offsets, always valid
#pragma unroll
                for (int n = 0; n < M_NG; ++n) {
                    xl[n] = *reinterpret_cast<const f32x4 *>(xp0[n] + el);
                    xh[n] = *reinterpret_cast<const f32x4 *>(xp0[n] + eh);
                }
            };
            f32x4 xl[M_NG], xh[M_NG];
            prefetch(0);
            loadx(0, xl, xh);
            for (int ks = 0; ks < NKS; ++ks) {
                uint32_t *buf = lds[ks & 1];
                // the buffer being overwritten was last read two k-steps ago; one barrier per k-step suffices
                commit(buf);
                __syncthreads();
                u32x4 xb[M_NG];
                {
                    const int e = 32 * ks + 8 * g;
                    const bool vl = e < A.ldx, vh = e + 4 < A.ldx;
                    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int n = 0; n < M_NG; ++n) xb[n] = pack_bf16x8((pv[n] && vl) ? xl[n] : zero, (pv[n] && vh) ? xh[n] : zero);
                }
                // unconditional (clamped): after the last k-step the loads re-read lines that are still in L2 and are never used
                prefetch(min(ks + 1, NKS - 1));
                loadx(min(ks + 1, NKS - 1), xl, xh);
#pragma unroll
                for (int rb = 0; rb < B_RBP; ++rb) {
                    if (rb < nrb) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const u32x4 a = *reinterpret_cast<const u32x4 *>(buf + (rb * 3 + pl) * 256 + lane * 4);
#pragma unroll
                            for (int n = 0; n < M_NG; ++n)
                                acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, xb[n]), acc[rb][n], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int rb = 0; rb < B_RBP; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * (rb0 + rb) + 4 * g + r;
                    if (rb < nrb && row < rows) {
                        const float cst = A.cst[row];
#pragma unroll
                        for (int n = 0; n < M_NG; ++n) scr[(int64_t)row * sstride + 16 * n + ci] = acc[rb][n][r] + cst;
                    }
                }
            __syncthreads();  // LDS buffers are reused by the next pass / tile
        }
        __syncthreads();
        const int64_t myp = wbase + lane;
        const bool valid = myp < A.n;
        if (valid && !A.labels_only) {
            const float *col = scr + lane;
            const Philox4 rr = philox4x32_10(A.seed, (uint64_t)(A.first_index + myp), A.epoch, STREAM_SWEEP);
            int z = 0;
            float m = -INFINITY;
            int best = 0;
            bool nan_seen = false;
            for (int k = 0; k < K; ++k) {
                const float a = col[(int64_t)(3 * k) * sstride];
                if (a != a) {
                    if (!nan_seen) { nan_seen = true; best = k; }
                } else if (a > m) {
                    m = a;
                    if (!nan_seen) best = k;
                }
            }
            if (A.final_argmax) {
                z = best;
            } else if (m == -INFINITY) {
                z = 0;
            } else {
                float s = 0.f;
                for (int k = 0; k < K; ++k) s += exp_det(nan_to_ninf(col[(int64_t)(3 * k) * sstride]) - m);
                const float t = u01(rr.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                for (int k = 0; k < K; ++k) {
                    cw += exp_det(nan_to_ninf(col[(int64_t)(3 * k) * sstride]) - m);
                    if (!(cw < t)) { z = k; break; }
                }
            }
            const float b0 = col[(int64_t)(3 * z + 1) * sstride], b1 = col[(int64_t)(3 * z + 2) * sstride];
            A.bins[myp] = 2 * z + draw2(b0, b1, u01(rr.v[1]));
        }
        __syncthreads();
    }
}

// Lp16[ks][rb][plane][lane][8 bf16] : element j of lane (i, g) = plane_p(logp[16 rb + i][32 ks + 8 g + j])
__global__ void mult_pack_bf16_kernel(const float *__restrict__ logp, uint32_t *__restrict__ Lp16, int rows, int64_t ldx, int NKS, int NRB) {
    const int64_t total = (int64_t)NKS * NRB * 3 * 256;   // uint32 words, two bf16 each
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(e & 3), lane = (int)((e >> 2) & 63);
        int64_t t = e >> 8;
        const int pl = (int)(t % 3); t /= 3;
        const int rb = (int)(t % NRB), ks = (int)(t / NRB);
        const int row = 16 * rb + (lane & 15);
        uint32_t out = 0;
        for (int h = 0; h < 2; ++h) {
            const int col = 32 * ks + 8 * (lane >> 4) + 2 * w + h;
            float v = (row < rows && col < ldx) ? logp[(size_t)row * ldx + col] : 0.f;
            uint32_t bits = 0;
            for (int p = 0; p <= pl; ++p) {
                bits = bf16_rne_bits(v);
                v -= __uint_as_float(bits << 16);
            }
            out |= (bits & 0xffffu) << (16 * h);
        }
        Lp16[e] = out;
    }
}

// Lp8[sl][vb][plane][lane][8 bf16], sl = 4 ks + q : element j of lane (i, g) = plane_p(logp[row(vb, i)][128 ks + 32 g + 8 q + j]) with the row
// blocks in two groups: vb < NRBc: the cluster row of cluster 16 vb + i (source row 3k); vb = NRBc + j: the left / right row (i odd) of
// cluster 8 j + i / 2 (source row 3k + 1 + i % 2) -- see mult_sweep_u8_kernel.  3K <= 16 (NRBs = 0): all rows in block 0, in source order
__global__ void mult_pack_u8_kernel(const float *__restrict__ logp, uint32_t *__restrict__ Lp8, int K, int64_t ldx, int NSL, int NRBc, int NRBs) {
    const int NRB = NRBc + NRBs;
    const int64_t total = (int64_t)NSL * NRB * 3 * 256;   // uint32 words, two bf16 each
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(e & 3), lane = (int)((e >> 2) & 63);
        int64_t t = e >> 8;
        const int pl = (int)(t % 3); t /= 3;
        const int vb = (int)(t % NRB), sl = (int)(t / NRB);
        const int i = lane & 15;
        const int k = NRBs == 0 ? i / 3 : (vb < NRBc ? 16 * vb + i : 8 * (vb - NRBc) + (i >> 1));
        const int row = NRBs == 0 ? i : 3 * k + (vb < NRBc ? 0 : 1 + (i & 1));
        uint32_t out = 0;
        for (int h = 0; h < 2; ++h) {
            const int col = 128 * (sl >> 2) + 32 * (lane >> 4) + 8 * (sl & 3) + 2 * w + h;
            float v = (k < K && col < ldx) ? logp[(size_t)row * ldx + col] : 0.f;
            uint32_t bits = 0;
            for (int p = 0; p <= pl; ++p) {
                bits = bf16_rne_bits(v);
                v -= __uint_as_float(bits << 16);
            }
            out |= (bits & 0xffffu) << (16 * h);
        }
        Lp8[e] = out;
    }
}

// upload: the byte copy of the points ([n][ld8], zero padded) + a flag that is raised when any element is not an integer in [0, 255]
__global__ void u8_convert_kernel(const float *__restrict__ X, int64_t ldx, int D, int64_t n, uint8_t *__restrict__ X8, int64_t ld8, int *__restrict__ flag) {
    int bad = 0;
    const int64_t total = n * ld8;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / ld8;
        const int d = (int)(e - i * ld8);
        uint8_t b = 0;
        if (d < D) {
            const float v = X[i * ldx + d];
            const int iv = (int)v;
            bad |= !(v >= 0.f && v <= 255.f && (float)iv == v);
            b = (uint8_t)(iv & 0xff);
        }
        X8[e] = b;
    }
    if (bad) atomicOr(flag, 1);
}

hipError_t launch_u8_convert(const float *X, int64_t ldx, int D, int64_t n, uint8_t *X8, int64_t ld8, int *d_flag, hipStream_t s) {
    DPMM_LAUNCH(u8_convert_kernel, dim3(4096), dim3(256), 0, s, X, ldx, D, n, X8, ld8, d_flag);
    return hipGetLastError();
}

// row blocks of the u8 planes for K clusters: cluster blocks, sub-cluster blocks (none when all 3K rows fit one block)
static inline void u8_blocks(int K, int &NRBc, int &NRBs) {
    NRBc = (K + 15) / 16;
    NRBs = 3 * K <= 16 ? 0 : (K + 7) / 8;
}

size_t mult_pack_u8_words(int rows, int64_t ld8) {
    int NRBc, NRBs;
    u8_blocks(rows / 3, NRBc, NRBs);
    const int NSL = (int)(ld8 / 32), NRB = NRBc + NRBs;
    return (size_t)NSL * NRB * 3 * 256;
}

hipError_t launch_mult_pack_u8(const float *logp, uint32_t *Lp8, int rows, int64_t ldx, int64_t ld8, hipStream_t s) {
    const int K = rows / 3, NSL = (int)(ld8 / 32);
    int NRBc, NRBs;
    u8_blocks(K, NRBc, NRBs);
    DPMM_LAUNCH(mult_pack_u8_kernel, dim3(512), dim3(256), 0, s, logp, Lp8, K, ldx, NSL, NRBc, NRBs);
    return hipGetLastError();
}

// dynamic LDS of mult_sweep_u8_kernel<B>: the two fragment buffers, or the tile's K x 256 cluster-level table (it aliases them) if that is larger.
// A compute unit has 160 KiB: the instantiations that run two workgroups per unit take up to 76 KiB each, the single-workgroup ones up to 152.
template <int B>
static hipError_t launch_u8(const MultSweepArgs &a, const uint8_t *X8, int64_t ld8, const uint32_t *Lp8, int grid, hipStream_t s) {
    const int NKS8 = (int)(ld8 / 128);
    int NRBc, NRBs;
    u8_blocks(a.K, NRBc, NRBs);
    const size_t frag = sizeof(uint32_t) * 2 * B * 3 * 256, tab = sizeof(float) * 256 * (size_t)a.K;
    const size_t cap = (B <= 4 ? 76 : 152) * 1024;
    const int ltab_ok = tab <= cap;
    size_t lds = ltab_ok && tab > frag ? tab : frag;
    if (ltab_ok && a.K <= 32 && lds < 32 * 1024) lds = 32 * 1024;      // the register draw reads 32 rows of the table whatever K is (rows past K are ignored)
    static size_t attr = 48 * 1024;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mult_sweep_u8_kernel<B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
        if (e != hipSuccess) return e;
        attr = cap;
    }
    DPMM_LAUNCH(mult_sweep_u8_kernel<B>, dim3(grid), dim3(256), lds, s, a, X8, ld8, Lp8, NKS8, NRBc, NRBs, ltab_ok);
    return hipGetLastError();
}

hipError_t launch_mult_sweep_u8(const MultSweepArgs &a, const uint8_t *X8, int64_t ld8, const uint32_t *Lp8, int grid, hipStream_t s) {
    // row blocks of a typical tile of a running chain: the cluster blocks + two sub-cluster blocks (table mode: all of them)
    int NRBc, NRBs;
    u8_blocks(a.K, NRBc, NRBs);
    // row blocks of a typical tile of a running chain: the cluster blocks + ONE sub-cluster block (a tile of the bin-sorted order that spans two
    // groups of eight clusters takes a second pass when that exceeds the instantiation); table mode: all of them
    const int typical = a.labels_only ? NRBc + NRBs : NRBc + (NRBs < 1 ? NRBs : 1);
    if (typical <= 2) return launch_u8<2>(a, X8, ld8, Lp8, grid, s);
    if (typical <= 4) return launch_u8<4>(a, X8, ld8, Lp8, grid, s);
    if (typical <= 6) return launch_u8<6>(a, X8, ld8, Lp8, grid, s);
    return launch_u8<8>(a, X8, ld8, Lp8, grid, s);
}

// data check at upload: 1 if every element is exactly representable in bf16 (low 16 mantissa bits zero)
__global__ void bf16_exact_kernel(const float *__restrict__ X, int64_t nwords, int *__restrict__ flag) {
    int bad = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x)
        bad |= (__float_as_uint(X[i]) & 0xffffu) != 0u;
    if (bad) atomicOr(flag, 1);
}

hipError_t launch_bf16_exact_check(const float *X, int64_t nwords, int *d_flag, hipStream_t s) {
    DPMM_LAUNCH(bf16_exact_kernel, dim3(2048), dim3(256), 0, s, X, nwords, d_flag);
    return hipGetLastError();
}

size_t mult_pack_bf16_words(int rows, int64_t ldx) {
    const int NKS = (int)((ldx + 31) / 32), NRB = (rows + 15) / 16;
    return (size_t)NKS * NRB * 3 * 256;
}

hipError_t launch_mult_pack_bf16(const float *logp, uint32_t *Lp16, int rows, int64_t ldx, hipStream_t s) {
    const int NKS = (int)((ldx + 31) / 32), NRB = (rows + 15) / 16;
    DPMM_LAUNCH(mult_pack_bf16_kernel, dim3(512), dim3(256), 0, s, logp, Lp16, rows, ldx, NKS, NRB);
    return hipGetLastError();
}

hipError_t launch_mult_sweep_bf16(const MultSweepArgs &a, const uint32_t *Lp16, int grid, hipStream_t s) {
    const int NKS = (int)((a.ldx + 31) / 32), NRB = (3 * a.K + 15) / 16;
    if (NRB <= 2) DPMM_LAUNCH(mult_sweep_bf16_kernel<2>, dim3(grid), dim3(256), 0, s, a, Lp16, NKS, NRB);
    else if (NRB <= 4) DPMM_LAUNCH(mult_sweep_bf16_kernel<4>, dim3(grid), dim3(256), 0, s, a, Lp16, NKS, NRB);
    else if (NRB <= 6) DPMM_LAUNCH(mult_sweep_bf16_kernel<6>, dim3(grid), dim3(256), 0, s, a, Lp16, NKS, NRB);
    else DPMM_LAUNCH(mult_sweep_bf16_kernel<8>, dim3(grid), dim3(256), 0, s, a, Lp16, NKS, NRB);
    return hipGetLastError();
}

int mult_tile_points() { return M_TILE; }

hipError_t launch_mult_pack(const float *logp, float *Lp, int rows, int64_t ldx, hipStream_t s) {
    const int NT = (int)((ldx + 15) / 16), NRB = (rows + 15) / 16;
    DPMM_LAUNCH(mult_pack_kernel, dim3(512), dim3(256), 0, s, logp, Lp, rows, ldx, NT, NRB);
    return hipGetLastError();
}

hipError_t launch_mult_sweep(const MultSweepArgs &a, int grid, hipStream_t s) {
    const int NT = (int)((a.ldx + 15) / 16), NRB = (3 * a.K + 15) / 16;
    DPMM_LAUNCH(mult_sweep_kernel, dim3(grid), dim3(256), 0, s, a, a.logp, NT, NRB);
    return hipGetLastError();
}

}  // namespace dpmm
