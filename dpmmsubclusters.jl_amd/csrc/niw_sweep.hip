// niw_sweep.hip -- fused label + sub-label sampling for the NIW / Gaussian prior on gfx950.
//
// Stands in for (reference paths relative to the reference checkout):
//   sample_labels_worker!            src/local_clusters_actions.jl:112-134
//   log_likelihood!(::mv_gaussian)   src/distributions/mv_gaussian.jl:21-25  (+ utils.jl:75-84)
//   sample_log_cat_array!            src/utils.jl:19-31
//   sample_sub_clusters_worker!      src/local_clusters_actions.jl:70-81
//   create_subclusters_labels!       src/local_clusters_actions.jl:83-95
//
// Design (CDNA4, wave64):
//   * the quadratic form is evaluated through the upper-triangular factor R of Sigma^-1
//     (Sigma^-1 = R'R):  q = || R (x - mu) ||^2 .  y = R z is a (D x D)(D x P) product on the
//     FP32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32 fma chains); only the
//     16x16 blocks on or above the diagonal are visited (D=64: 10 of 16 blocks).
//   * one wave owns 16*NG points.  Their x stays in registers for the whole sweep (the
//     B operand); z = x - mu_k is formed on the fly; the A operand is a pre-packed fragment
//     image of R_k, 1 KiB per 16x16 block.
//   * TWO kernels share this file:
//       niw_sweep_direct_kernel (D <= 64; the benchmark's kernel): no LDS staging, no barriers -- every wave streams the
//         fragments of a matrix from L2 into registers one row-block ahead of its MFMAs, keeps its a_k table in
//         wave-private LDS, screens clusters per WAVE (reference evaluation -> 4-row tail screen on the VALU -> 16-row
//         MFMA screen -> full evaluation of survivors) and walks the distinct labels of its own 64 points for the
//         sub-labels.  See the comment block in front of it.
//       niw_sweep_kernel (D = 128, 256): the fragment image of a matrix is staged global -> registers -> LDS one
//         chunk ahead of the MFMAs and shared by the 4 waves of the workgroup; screening is workgroup-wide; a_k goes to a
//         per-workgroup scratch row (L2-resident).
//   * a_k = -1/2 q + (-1/2 logdet_k + log w_k); each lane then draws the label of "its" point with the
//     deterministic inverse-CDF scan shared with the CPU oracle, using Philox(seed; global index, epoch).
//   * sub-labels: for each distinct label just drawn, the left/right sub-cluster forms are evaluated for the
//     points that carry it and 1 of 2 is drawn with the second uniform.
//
// Fragment conventions for v_mfma_f32_16x16x4_f32 (lane l: i = l & 15, g = l >> 4):
//   A[i][k=g], B[k=g][col=i], C/D reg r: row 4g + r, col i.
// Contraction step (t, jj) covers vector elements {16t + 4g + jj : g = 0..3}; x is loaded as
// float4 at element 16t + 4g, so element (16t + 4g + jj) is component jj of that float4.
#include "dpmm_device.h"
#include "dpmm_kernels.h"
#include "niw_device.h"
#include "niw_b3.h"
#include <cstdlib>
#include <cstring>

namespace dpmm {

template <int NB, int NG, int CH>
struct NiwCfg {
    static constexpr int DP = 16 * NB;
    static constexpr int NP = NB * (NB + 1) / 2;     // 16x16 blocks on/above the diagonal
    static constexpr int MATSZ = NP * 256;            // floats per packed matrix
    // Chunks = runs of whole row blocks holding at most CAP 16x16 blocks, cut greedily from the top (row block bi has NB - bi of
    // them): D = 256 -> 5 chunks of 31 / 27 / 23 / 27 / 28 blocks, D = 128 -> 3 chunks of 15 / 15 / 6.  (One row block per chunk, as
    // before, ended in chunks of 3, 2, 1 blocks: 8-24 matrix instructions between two barriers, with the global -> register latency
    // of the next chunk fully exposed.)  CH is kept as a tag of the configuration.
    static constexpr int CAP = NB >= 16 ? 32 : 16;
    __host__ __device__ static constexpr int chunk_row(int c) {     // first row block of chunk c (c == NCH: NB)
        int r = 0;
        for (int i = 0; i < c; ++i) {
            int p = 0;
            while (r < NB && p + (NB - r) <= CAP) { p += NB - r; ++r; }
        }
        return r;
    }
    __host__ __device__ static constexpr int count_chunks() {
        int c = 0;
        while (chunk_row(c) < NB) ++c;
        return c;
    }
    static constexpr int NCH = count_chunks();        // chunks per matrix
    static constexpr int TILE = 64 * NG;              // points per workgroup tile (4 waves x 16 NG)
    static constexpr int WPTS = 16 * NG;              // points per wave
    __host__ __device__ static constexpr int row0(int c) { return chunk_row(c); }
    __host__ __device__ static constexpr int row1(int c) { return chunk_row(c + 1); }
    __host__ __device__ static constexpr int pairs(int c) { return pair_base<NB>(row1(c)) - pair_base<NB>(row0(c)); }
    __host__ __device__ static constexpr int passes(int c) { return (pairs(c) + 3) / 4; }
    __host__ __device__ static constexpr int max_pairs() {
        int m = 0;
        for (int c = 0; c < NCH; ++c) m = pairs(c) > m ? pairs(c) : m;
        return m;
    }
    static constexpr int MAXPAIRS = max_pairs();
    static constexpr int MAXPASS = (MAXPAIRS + 3) / 4;
};

template <int NG>
__device__ __forceinline__ void ref_bracket(const uint32_t *__restrict__ Rb, const f32x4 (&x)[NG][4], const f32x4 (&mu)[4], int lane, float (&qhi)[NG],
                                            const float refb_c = REFB_C) {
    static_assert(NG % 2 == 0, "point groups are taken two at a time");
    const u32x4_t *F = reinterpret_cast<const u32x4_t *>(Rb) + lane;          // fragment f of this lane: F[64 f]
    const u32x4_t absm = (u32x4_t){0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu};
    // the six fragments are requested first (their L2 latency runs under the conversion of z) and serve both pairs of point groups (held
    // from the top of the tile, under the gather of x, they cost 52 spilled registers);
    // two point groups at a time: the bf16 operands of all four groups plus both accumulator sets are registers the kernel does not have
    u32x4_t a[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) a[f] = F[64 * f];
#pragma unroll
    for (int n0 = 0; n0 < NG; n0 += 2) {
        u32x4_t zb[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const f32x4 lo = x[n0 + h][2 * sl] - mu[2 * sl], hi = x[n0 + h][2 * sl + 1] - mu[2 * sl + 1];
                zb[h][sl] = (u32x4_t){pack_bf16_pair(lo.x, lo.y), pack_bf16_pair(lo.z, lo.w), pack_bf16_pair(hi.x, hi.y), pack_bf16_pair(hi.z, hi.w)};
            }
        float part[2] = {0.f, 0.f};
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
            f32x4 y[2], e[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) { y[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; e[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int f = bi == 0 ? sl : (bi == 1 ? 2 + sl : (sl == 1 ? bi + 2 : -1));
                if (f < 0) continue;
                const u32x4_t aa = a[f] & absm;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    y[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[f]), __builtin_bit_cast(bf16x8_t, zb[h][sl]), y[h], 0, 0, 0);
                    e[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aa), __builtin_bit_cast(bf16x8_t, zb[h][sl] & absm), e[h], 0, 0, 0);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = __builtin_fmaf(refb_c, e[h][r], fabsf(y[h][r]));
                    part[h] = __builtin_fmaf(t, t, part[h]);
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 tot = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, part[h], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);      // sum over the four row groups of a column
            qhi[n0 + h] = __builtin_fmaf(tot[0], 1.0001f, 1e-20f);
        }
    }
}

// bf16 SCREENS (D in 33 .. 64, DPMM_OPT_BF16_SCREENS): the same certified bound as the bracket, the other way round.  For the rows of a
// block row of y = R z,  |y_r| >= |y^_r| - REFB_C e^_r,  so  q >= sum_r max(0, |y^_r| - REFB_C e^_r)^2  is a LOWER bound of the quadratic form from
// two bf16 matrix passes -- and cst - q_lb / 2 an upper bound of a_k, the only thing a screen needs.  Both screens sit IN FRONT of the
// Float32 test they imitate and only ever skip it: a cluster they exclude would have been excluded by the Float32 test as well (their
// bound is the weaker one), a cluster they let through takes the Float32 test as before.  The set of evaluated clusters, the table and
// the labels are those of the kernel without them, bit for bit.
//   bottom: rows 48 .. 63 (block row 3: the last 16 features only) -- fragment 5 of the cluster's bf16 image; 8 matrix instructions of
//           16 cycles against the Float32 screen's 16 of 32; per lane its own four rows, as the Float32 screen;
//   top:    rows 0 .. 15 (block row 0: all 64 features) -- fragments 0, 1; 16 matrix instructions + 4 row sums against the 64 + 4 Float32
//           ones of the evaluation's first row block (quad_stream<.., EARLY>); for the clusters the bottom screen lets through, still in
//           front of their Float32 16-row screen (every exclusion here is one the Float32 tests would have made: the order is free).
// Overlapping clusters (component means at MixtureVar 4 / 1 instead of 100) are where they pay: there the 4-row bounds exclude nothing
// and a tile runs ~30 sixteen-row screens (MixtureVar 4), or ~19 evaluations that leave after their first row block (MixtureVar 1).
template <int NG>
__device__ __forceinline__ bool bf16_top_excludes(const uint32_t *__restrict__ Rb, const f32x4 (&x)[NG][4], const f32x4 (&mu)[4], int lane, float cst,
                                                  const float (&thr)[NG]) {
    static_assert(NG % 2 == 0, "point groups are taken two at a time");
    const u32x4_t absm = (u32x4_t){0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu};
    const u32x4_t a0 = reinterpret_cast<const u32x4_t *>(Rb)[lane], a1 = reinterpret_cast<const u32x4_t *>(Rb)[64 + lane];      // block row 0 x features 0 .. 31 | 32 .. 63
    const u32x4_t aa0 = a0 & absm, aa1 = a1 & absm;
    bool out = true;
#pragma unroll
    for (int n0 = 0; n0 < NG; n0 += 2) {
        float part[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4_t zb[2];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const f32x4 lo = x[n0 + h][2 * sl] - mu[2 * sl], hi = x[n0 + h][2 * sl + 1] - mu[2 * sl + 1];
                zb[sl] = (u32x4_t){pack_bf16_pair(lo.x, lo.y), pack_bf16_pair(lo.z, lo.w), pack_bf16_pair(hi.x, hi.y), pack_bf16_pair(hi.z, hi.w)};
            }
            f32x4 y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, zb[0]), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aa0), __builtin_bit_cast(bf16x8_t, zb[0] & absm), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, zb[1]), y, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aa1), __builtin_bit_cast(bf16x8_t, zb[1] & absm), e, 0, 0, 0);
            float ql = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = fmaxf(__builtin_fmaf(-REFB_C, e[r], fabsf(y[r])), 0.f);
                ql = __builtin_fmaf(t, t, ql);
            }
            part[h] = ql;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 tot = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, part[h], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);      // sum over the four row groups of a column
            out = out && (__builtin_fmaf(-0.5f * 0.9999f, tot[0], cst) < thr[n0 + h]);
        }
    }
    return __all(out);
}

// DIRECTION screen (D in 33 .. 64, K <= 64; DPMM_OPT_DIRECTION_SCREEN): every remaining candidate of a tile (six or more) at once, each along the ONE
// direction that separates it from the wave's reference cluster k0.  For a unit vector u,  q_k(x) = |R_k (x - mu_k)|^2 >= (u' R_k (x - mu_k))^2;
// with d = mu_k0 - mu_k, b = |R_k d|, u = R_k d / b and w = R_k' u (launch_niw_direction: one 64-vector and two scalars per ordered pair)
//     u' R_k (x - mu_k) = w . z0 + b,      z0 = x - mu_k0,
// b is the Mahalanobis distance of the reference cluster's mean from cluster k and w . z0 the point's own offset along that direction -- for
// a point of k0 a few of ITS standard deviations.  The K dot products of a point are ONE bf16 matrix product per 16 clusters and 32 features,
// w . z0 = s^ +- e |z0| with e = c |w| (the bracket's rounding constant c: sum |a_i||b_i| <= |a||b|), so
//     a_k(x) <= cst_k - 0.4995 t^2,   t = |s^ + b| - e |z0|,  counted only when t >= 2 % of b
// (the Float32 roundings of b, of |u| = 1 and of d are a few 1e-6 of b: below the 1e-3 taken off t^2 once t is a percent of b; a test that
// decides anything has t^2 / 2 > 50 nats).
// The reference generator's clusters (covariances ~ InvWishart(D + 2, I): condition numbers of 1e4) are separated by hundreds of such units
// at MixtureVar 4 and 1 although no 4-feature bound separates them: there this one test removes what took ~30 sixteen-row screens (and ~20
// first-row-block screens) per tile.  It only ever removes candidates, each exclusion is one the Float32 evaluation would have made, so the
// table, the evaluated set and the labels are those of the kernel without it.  Returns the mask of excluded clusters (bit k).
// 16 (K <= 32) or 32 bf16 matrix instructions + 4 row sums per tile, ~400 vector instructions.
template <int NG>
__device__ __forceinline__ unsigned long long direction_far(const uint32_t *__restrict__ frag, const float *__restrict__ cons, const float *__restrict__ mup0,
                                                            const f32x4 (&x)[NG][4], const float (&thr)[NG], int lane, int g, int K) {
    f32x4 m0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) m0[t] = *reinterpret_cast<const f32x4 *>(mup0 + 16 * t + 4 * g);
    u32x4_t zb[NG][2];
    float nz[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        float part = 0.f;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const f32x4 lo = x[n][2 * sl] - m0[2 * sl], hi = x[n][2 * sl + 1] - m0[2 * sl + 1];
            part = __builtin_fmaf(lo.x, lo.x, part); part = __builtin_fmaf(lo.y, lo.y, part); part = __builtin_fmaf(lo.z, lo.z, part); part = __builtin_fmaf(lo.w, lo.w, part);
            part = __builtin_fmaf(hi.x, hi.x, part); part = __builtin_fmaf(hi.y, hi.y, part); part = __builtin_fmaf(hi.z, hi.z, part); part = __builtin_fmaf(hi.w, hi.w, part);
            zb[n][sl] = (u32x4_t){pack_bf16_pair(lo.x, lo.y), pack_bf16_pair(lo.z, lo.w), pack_bf16_pair(hi.x, hi.y), pack_bf16_pair(hi.z, hi.w)};
        }
        nz[n] = direction_norm(part);
    }
    return direction_far_core<NG>(frag, cons, zb, nz, thr, lane, g, K);      // (niw_device.h: shared with niw_lean_kernel, whose plane h of z0 IS zb)
}

// Reference bracket of the LDS-staged kernels (D = 128, 256: NB = 8, 16): the same certified bound as ref_bracket -- q <= sum_r (|y^_r| + c e^_r)^2
// from two bf16 matrix passes, y^ = R~ z~, e^ = |R~||z~| -- over the bf16 image of the cluster-level factor: fragments (block row bi, 32-feature
// slice sl >= bi / 2) in that order, each [64 lanes][4 dwords] with the k-slots of refb_map (launch_niw_refb_big converts the Float32
// fragment image; NB = 16: 72 fragments = 72 KiB per cluster against the 136 KiB of the Float32 image).  Every wave streams the fragments
// itself from L2 (no staging, no barrier: 4 x 72 KiB per 128-point tile) one block row ahead of the matrix instructions: 2 x 72 x NG
// instructions of 16 cycles against the evaluation's 136 x 4 x NG of 32.  c: the bracket's constant with the accumulation term of up to
// 256 products per row.
constexpr float REFB_C_BIG = 0.00790f;       // 2^-7 (1 + 2^-8) + 3 * 256 * 2^-24 = 0.0078736
template <int NB>
__host__ __device__ constexpr int refb_big_frags() { int c = 0; for (int bi = 0; bi < NB; ++bi) c += NB / 2 - bi / 2; return c; }
// The bracket of the LDS-staged kernels as a launch of its own, one workgroup per 128-point tile of the sweep (same tiles, same visiting
// order): k0 = the previous label of the tile's first point; if EVERY point of the tile had label k0, the bracket's lower end of a_k0 for
// every point -> aref[position], tile_flag[tile] = 1 + k0; else tile_flag[tile] = 0 (the sweep evaluates its references as before).
// Inside the sweep kernel the same arithmetic cost more than it saved at D = 256 (178 spilled registers).  Here a wave converts its 32
// points to the bf16 operands z~ (all the registers it keeps: x itself is dropped) and the workgroup shares the fragments through LDS --
// chunks of CHF fragments, two buffers, the next chunk requested before this one's matrix instructions -- so a tile reads its 72 KiB image
// once (every wave streaming it for itself: 1.4 GB from L2 per launch at the C5 shard, 0.35 ms).  It reads X a second time (0.64 GB).
template <int NB, int NG>
__global__ __launch_bounds__(256) void niw_bracket_big_kernel(const float *__restrict__ X, int64_t ldx, int64_t n, const int32_t *__restrict__ order,
                                                              const int32_t *__restrict__ order_total, const int32_t *__restrict__ bins, int K,
                                                              const float *__restrict__ mup, const float *__restrict__ cst,
                                                              const uint32_t *__restrict__ refb, uint32_t *__restrict__ tile_flag, float *__restrict__ aref) {
    constexpr int WPTS = 16 * NG, TILE = 4 * WPTS, DP = 16 * NB, NSL = NB / 2, NF = refb_big_frags<NB>();
    constexpr int CHF = 4;                                   // fragments per chunk: 4 KiB = one 16-byte vector per thread
    static_assert(NF % CHF == 0, "whole chunks");
    __shared__ __attribute__((aligned(16))) uint32_t fbuf[2][CHF * 256];
    __shared__ int sh_first[4], sh_bad[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ci = lane & 15, g = lane >> 4;
    const int64_t tile = blockIdx.x;
    const bool use_order = order != nullptr && *order_total == (int32_t)n;
    const int64_t wbase = tile * TILE + (int64_t)wave * WPTS, mypos = wbase + lane;
    const bool valid = lane < WPTS && mypos < n;
    const int64_t myp = (valid && use_order) ? (int64_t)order[mypos] : mypos;
    int prev = valid ? (bins[myp] >> 1) : -1;
    if ((unsigned)prev >= (unsigned)K) prev = -1;
    {
        const unsigned long long pm = __ballot(prev >= 0);
        const int kf = pm ? __shfl(prev, __ffsll((long long)pm) - 1) : -1;
        if (lane == 0) sh_first[wave] = kf;
    }
    __syncthreads();
    int k0 = 0;
    for (int wv = 3; wv >= 0; --wv) if (sh_first[wv] >= 0) k0 = sh_first[wv];
    {
        const bool bad = __ballot(valid && prev != k0) != 0ull;
        if (lane == 0) sh_bad[wave] = bad ? 1 : 0;
    }
    __syncthreads();
    const bool homog = (sh_bad[0] | sh_bad[1] | sh_bad[2] | sh_bad[3]) == 0 && (sh_first[0] >= 0 || sh_first[1] >= 0 || sh_first[2] >= 0 || sh_first[3] >= 0);
    if (tid == 0) tile_flag[tile] = homog ? (uint32_t)(k0 + 1) : 0u;
    if (!homog) return;                                      // (workgroup-uniform)
    // the image of k0: chunk c = fragments CHF c .. CHF c + CHF - 1, thread tid moves vector tid of the chunk's 256
    const u32x4_t *img = reinterpret_cast<const u32x4_t *>(refb + (size_t)k0 * (NF * 256));
    constexpr int NCHK = NF / CHF, AHEAD = 4;                // chunks requested ahead of the one being multiplied (registers: one vector each)
    u32x4_t st[AHEAD];
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) st[q] = img[(size_t)(q < NCHK ? q : NCHK - 1) * 256 + tid];      // chunks 0 .. AHEAD - 1, under the gather of x
    // z~ = bf16(x - mu_k0) of this wave's points, slice by slice (x is not kept)
    const float *mu0 = mup + (size_t)(3 * k0) * DP;
    u32x4_t zb[NG][NSL];
    {
        int64_t prow[NG];
        bool pok[NG];
#pragma unroll
        for (int nn = 0; nn < NG; ++nn) {
            const int64_t pos = wbase + 16 * nn + ci;
            pok[nn] = pos < n;
            prow[nn] = (pok[nn] && use_order) ? (int64_t)order[pos] : pos;
        }
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            const int e0 = 32 * sl + 4 * g, e1 = e0 + 16;
            const f32x4 m0 = *reinterpret_cast<const f32x4 *>(mu0 + e0), m1 = *reinterpret_cast<const f32x4 *>(mu0 + e1);
#pragma unroll
            for (int nn = 0; nn < NG; ++nn) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 x0 = (pok[nn] && e0 < ldx) ? *reinterpret_cast<const f32x4 *>(X + prow[nn] * ldx + e0) : zero;
                const f32x4 x1 = (pok[nn] && e1 < ldx) ? *reinterpret_cast<const f32x4 *>(X + prow[nn] * ldx + e1) : zero;
                const f32x4 lo = x0 - m0, hi = x1 - m1;
                zb[nn][sl] = (u32x4_t){pack_bf16_pair(lo.x, lo.y), pack_bf16_pair(lo.z, lo.w), pack_bf16_pair(hi.x, hi.y), pack_bf16_pair(hi.z, hi.w)};
            }
        }
    }
    const u32x4_t absm = (u32x4_t){0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu};
    float part[NG];
    f32x4 y[NG], e[NG];
#pragma unroll
    for (int nn = 0; nn < NG; ++nn) { part[nn] = 0.f; y[nn] = (f32x4){0.f, 0.f, 0.f, 0.f}; e[nn] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    reinterpret_cast<u32x4_t *>(fbuf[0])[tid] = st[0];
    if (AHEAD < NCHK) st[0] = img[(size_t)AHEAD * 256 + tid];
    __syncthreads();
    // fragment f = (block row bi, slice sl >= bi / 2): everything below is unrolled, (bi, sl) are constants per fragment
    int fbase = 0, f = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        const int s0 = bi / 2, cnt = NSL - s0;
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            if (i < cnt) {
                const int c = f / CHF, j = f % CHF;
                const u32x4_t a = reinterpret_cast<const u32x4_t *>(fbuf[c & 1])[j * 64 + lane], aa = a & absm;
#pragma unroll
                for (int nn = 0; nn < NG; ++nn) {
                    y[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, zb[nn][s0 + i]), y[nn], 0, 0, 0);
                    e[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aa), __builtin_bit_cast(bf16x8_t, zb[nn][s0 + i] & absm), e[nn], 0, 0, 0);
                }
                if (j == CHF - 1 && c + 1 < NCHK) {           // end of the chunk: the next one goes to the other buffer (last read a chunk ago) ...
                    reinterpret_cast<u32x4_t *>(fbuf[(c + 1) & 1])[tid] = st[(c + 1) % AHEAD];
                    if (c + 1 + AHEAD < NCHK) st[(c + 1) % AHEAD] = img[(size_t)(c + 1 + AHEAD) * 256 + tid];      // ... and its register takes chunk c + 1 + AHEAD
                    __syncthreads();
                }
                ++f;
            }
        }
#pragma unroll
        for (int nn = 0; nn < NG; ++nn) {                                       // the block row's 16 rows are complete
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = __builtin_fmaf(REFB_C_BIG, e[nn][r], fabsf(y[nn][r]));
                part[nn] = __builtin_fmaf(t, t, part[nn]);
            }
            y[nn] = (f32x4){0.f, 0.f, 0.f, 0.f}; e[nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        fbase += cnt;
    }
    (void)fbase;
    float sel = 0.f;
#pragma unroll
    for (int nn = 0; nn < NG; ++nn) {                                           // sum over the four row groups of a column
        float v = part[nn];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (g == nn) sel = __builtin_fmaf(v, 1.0001f, 1e-20f);                 // owner lane 16 n + ci: point (n, ci)
    }
    if (valid) aref[mypos] = __builtin_fmaf(-0.5f, sel, cst[3 * k0]);
}

template <int NB, int NG, int CH>
struct QuadEval {
    using C = NiwCfg<NB, NG, CH>;
    // staging registers: named members + compile-time selector (an indexed array here ends up in
    // scratch memory: the conditional writes defeat SROA)
    f32x4 st0, st1, st2, st3, st4, st5, st6, st7;
    int par = 0;       // LDS buffer the NEXT commit goes to (two buffers, one barrier per chunk; runs on across matrices)
    static_assert(C::MAXPASS <= 8, "chunk too large for the staging registers");
    template <int p>
    __device__ __forceinline__ f32x4 &S() {
        if constexpr (p == 0) return st0;
        else if constexpr (p == 1) return st1;
        else if constexpr (p == 2) return st2;
        else if constexpr (p == 3) return st3;
        else if constexpr (p == 4) return st4;
        else if constexpr (p == 5) return st5;
        else if constexpr (p == 6) return st6;
        else return st7;
    }
    template <int c, int p>
    __device__ __forceinline__ void prefetch_pass(const f32x4 *src) {
        if constexpr (p < C::passes(c)) {
            const int i4 = p * 256 + (int)threadIdx.x;
            if constexpr ((p + 1) * 256 <= C::pairs(c) * 64) S<p>() = src[i4];
            else if (i4 < C::pairs(c) * 64) S<p>() = src[i4];
            prefetch_pass<c, p + 1>(src);
        }
    }
    template <int c, int p>
    __device__ __forceinline__ void commit_pass(f32x4 *dst) {
        if constexpr (p < C::passes(c)) {
            const int i4 = p * 256 + (int)threadIdx.x;
            if constexpr ((p + 1) * 256 <= C::pairs(c) * 64) dst[i4] = S<p>();
            else if (i4 < C::pairs(c) * 64) dst[i4] = S<p>();
            commit_pass<c, p + 1>(dst);
        }
    }
    // issue the global loads of chunk c of packed matrix Rm into registers
    template <int c>
    __device__ __forceinline__ void prefetch(const float *__restrict__ Rm) {
        prefetch_pass<c, 0>(reinterpret_cast<const f32x4 *>(Rm + pair_base<NB>(C::row0(c)) * 256));
    }
    template <int c>
    __device__ __forceinline__ void commit(float *lds) {
        commit_pass<c, 0>(reinterpret_cast<f32x4 *>(lds));
    }
    // accumulate q[n] += sum over the chunk's rows of y_row^2, y = R (x - mu)
    template <int c>
    __device__ __forceinline__ void compute(const float *lds, const f32x4 (&x)[NG][NB], const f32x4 (&mu)[NB],
                                            float (&q)[NG], int lane) {
        // the A fragment of the NEXT block is requested before the matrix instructions of the current one (the chunk's blocks are
        // contiguous in LDS in visiting order); scheduling fences keep the read there -- left alone, the compiler issues every
        // ds_read right in front of its first use and the LDS latency (~120 cycles per 8 matrix instructions at one wave per SIMD)
        // is fully exposed
        constexpr int P = C::pairs(c);
        int pic = 0;
        f32x4 a = *reinterpret_cast<const f32x4 *>(lds + lane * 4);
#pragma unroll
        for (int bi = C::row0(c); bi < C::row1(c); ++bi) {
            f32x4 acc[NG];
#pragma unroll
            for (int n = 0; n < NG; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = bi; t < NB; ++t) {
                f32x4 an = a;
                if (pic + 1 < P) an = *reinterpret_cast<const f32x4 *>(lds + (pic + 1) * 256 + lane * 4);
                __builtin_amdgcn_sched_barrier(0);
                // rotate over the NG independent accumulators: a dependent MFMA pair needs 40 cycles,
                // the issue interval is 32
                f32x4 zz[NG];
#pragma unroll
                for (int n = 0; n < NG; ++n) zz[n] = x[n][t] - mu[t];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                    for (int n = 0; n < NG; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], zz[n][jj], acc[n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                a = an;
                ++pic;
            }
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                q[n] = __builtin_fmaf(acc[n][0], acc[n][0], q[n]);
                q[n] = __builtin_fmaf(acc[n][1], acc[n][1], q[n]);
                q[n] = __builtin_fmaf(acc[n][2], acc[n][2], q[n]);
                q[n] = __builtin_fmaf(acc[n][3], acc[n][3], q[n]);
            }
        }
    }
};

// Evaluate q for one packed matrix (all chunks), prefetching the first chunk of `Rnext`
// (may be null) behind the last chunk's MFMAs.  On entry the first chunk of Rcur must already
// be in ev.st.  `active` is wave-uniform: inactive waves take part in the staging but skip the MFMAs.
template <int NB, int NG, int CH, int c = 0>
__device__ __forceinline__ void eval_matrix(QuadEval<NB, NG, CH> &ev, float *lds, const float *Rcur, const float *Rnext,
                                            const f32x4 (&x)[NG][NB], const f32x4 (&mu)[NB], float (&q)[NG],
                                            int lane, bool active) {
    using C = NiwCfg<NB, NG, CH>;
    // Two LDS buffers: the one written here was last read two chunks ago, and every wave has passed a barrier since it finished
    // that chunk -- ONE barrier per chunk (commit -> barrier -> prefetch the next chunk -> matrix instructions).
    float *buf = lds + ev.par * (C::MAXPAIRS * 256);
    ev.par ^= 1;
    ev.template commit<c>(buf);
    __syncthreads();
    if constexpr (c + 1 < C::NCH) {
        ev.template prefetch<c + 1>(Rcur);
    } else {
        if (Rnext) ev.template prefetch<0>(Rnext);
    }
    if (active) ev.template compute<c>(buf, x, mu, q, lane);
    if constexpr (c + 1 < C::NCH) eval_matrix<NB, NG, CH, c + 1>(ev, lds, Rcur, Rnext, x, mu, q, lane, active);
}

template <int NG>
__device__ __forceinline__ float reduce_select(float (&q)[NG], int g) {
    float sel = 0.f;
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        float v = q[n];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (g == n) sel = v;
    }
    return sel;
}

#ifdef DPMM_STAMPS
#define STAMP(var) unsigned long long var; do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var)
#endif
template <int NB, int NG, int CH>
__global__ __launch_bounds__(256, (NB <= 8 ? 2 : 1)) void niw_sweep_kernel(NiwSweepArgs A) {
#ifdef DPMM_STAMPS
    unsigned long long T_x = 0, T_ref = 0, T_tail = 0, T_surv = 0, T_draw = 0, T_p2 = 0, T_tot = 0, T_max = 0, T_first = 0, T_last = 0; int ntile = 0;
#endif
    using C = NiwCfg<NB, NG, CH>;
    __shared__ __attribute__((aligned(16))) float lds[2 * C::MAXPAIRS * 256];
    __shared__ uint32_t present[DPMM_MAX_CLUSTERS_K / 32];
    __shared__ uint32_t survm[DPMM_MAX_CLUSTERS_K / 32];   // screened mode: clusters some wave of the workgroup could not exclude
    __shared__ uint32_t surv2[DPMM_MAX_CLUSTERS_K / 32];   // ... and still could not after the 16-row screen
    __shared__ int sh_first[4], sh_last[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ci = lane & 15;  // column (point within a 16-group) for the B operand
    const int g = lane >> 4;   // k-slot / row group
    const int K = A.K;
    const bool use_order = A.order != nullptr && *A.order_total == (int32_t)A.n;
    const bool owner = lane < C::WPTS;  // lane `lane` draws for point `lane` of the wave

    QuadEval<NB, NG, CH> ev;
    unsigned nw_tiles = 0, nw_full = 0, nw_tail = 0, nw_scr = 0, nw_br = 0;   // executed-work counters of this wave (wave-uniform)

    // Tiles are handed out through a queue (A.work[4], cleared with the work counters): the cost of a tile varies with the number of
    // clusters its points cannot exclude (1 to > 10 full evaluations at D = 256), and with a static "tile = workgroup + i * grid"
    // assignment the slowest workgroup ran 1.9 x the median (2.76 ms kernel, 1.45 ms median workgroup).  Results do not depend on
    // which workgroup takes a tile (the random stream is keyed by the point).  Without counters (table mode): static assignment.
    // The claim of a tile is issued a tile AHEAD (an atomic with return, in flight during the current tile), and once it has arrived the
    // workgroup fetches the next tile's entries of the visiting order and previous labels: a tile then starts with its X gather instead
    // of the chain claim -> order -> X (7.6 % of the D = 256 launch sat in that phase at one wave per SIMD).
    __shared__ long long sh_tile, sh_next;
    const bool queue = A.work != nullptr;
    int64_t tile = blockIdx.x;
    long long pend = 0;                                     // thread 0: the claim in flight
    if (queue && tid == 0) pend = (long long)atomicAdd(&A.work[4], 1ull);      // in order: a strided order was measured 4 % slower
    int64_t nx_tile = -1;                                   // tile whose order / previous-label entries sit in pf_*
    int pf_ord[NG], pf_my = -1, pf_bin = -1;
    for (;; tile += gridDim.x) {
        if (queue) {
            __syncthreads();                               // everybody has read the previous value
            if (tid == 0) sh_tile = pend;
            __syncthreads();
            tile = sh_tile;
            if (tid == 0 && tile < A.ntiles) pend = (long long)atomicAdd(&A.work[4], 1ull);
        }
        if (tile >= A.ntiles) break;
        ++nw_tiles;
        STAMP(s0);
        const int64_t wbase = tile * C::TILE + (int64_t)wave * C::WPTS;  // first point of this wave
        const bool prefetched = queue && nx_tile == tile;
        // ---- x tile -> registers (B-operand layout)
        f32x4 x[NG][NB];
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            const int64_t pos = wbase + 16 * n + ci;
            const int64_t p = (pos < A.n && use_order) ? (prefetched ? (int64_t)pf_ord[n] : (int64_t)A.order[pos]) : pos;
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                const int e = 16 * t + 4 * g;
                x[n][t] = (pos < A.n && e < A.ldx) ? *reinterpret_cast<const f32x4 *>(A.X + p * A.ldx + e)
                                                    : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        const int64_t mypos = wbase + lane;  // owner's position in processing order
        const bool valid = owner && mypos < A.n;
        const int64_t myp = (valid && use_order) ? (prefetched ? (int64_t)pf_my : (int64_t)A.order[mypos]) : mypos;   // owner's point
        const int my_prev_bin = prefetched ? pf_bin : -2;     // (-2: not fetched)
#ifdef DPMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(s1);
        float *scr = A.scratch + (A.scratch_by_tile ? tile * C::TILE : (int64_t)blockIdx.x * C::TILE) + wave * C::WPTS + lane;
        const int64_t sstride = A.scratch_stride;

        // ---- phase 1: a_k for every cluster (or, screened, for the clusters that matter to this workgroup)
        float m_run = -INFINITY;
        int best = 0;
        bool nan_seen = false;
        auto eval_cluster = [&](int k, const float *Rnext) {
            f32x4 mu[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t)
                mu[t] = *reinterpret_cast<const f32x4 *>(A.mup + (size_t)(3 * k) * C::DP + 16 * t + 4 * g);
            float q[NG];
#pragma unroll
            for (int n = 0; n < NG; ++n) q[n] = 0.f;
            eval_matrix<NB, NG, CH>(ev, lds, A.Rp + (size_t)(3 * k) * C::MATSZ, Rnext, x, mu, q, lane, true);
            ++nw_full;
            const float qs = reduce_select<NG>(q, g);
            const float a = A.tdf ? A.cst[3 * k] - A.tdf[6 * k + 1] * log1pf(qs / A.tdf[6 * k])
                                  : __builtin_fmaf(-0.5f, qs, A.cst[3 * k]);
            if (valid) {
                scr[(int64_t)k * sstride] = a;
                if (a != a) {
                    if (!nan_seen || k < best) { nan_seen = true; best = k; }  // Julia argmax: first NaN wins
                } else if (a > m_run || (a == m_run && k < best && !nan_seen && m_run != -INFINITY)) {
                    m_run = a;
                    if (!nan_seen) best = k;
                }
            }
            return a;
        };
        const bool screening = A.tail != nullptr && A.screen_margin > 0.f && !A.tdf && !A.scratch_by_tile && K > 1;
        int k0 = 0, k1 = 0;
        bool ref_skipped_tile = false;          // bracketed reference and no survivor (workgroup-uniform): k0's value was never computed
#ifdef DPMM_STAMPS
        unsigned long long sv0 = s1;
#endif
        // (ONE cluster, no table asked for: the draw returns index 0 whatever its value -- utils.jl:19-31 over one element -- so the value is not
        // computed: every fit from init_clusters = 1 spends its first `burnout` sweeps here, two evaluations per tile instead of three)
        const bool lone = K == 1 && !A.tdf && !A.labels_only && !A.scratch_by_tile;
        if (!screening) {
            if (!lone) {
            ev.template prefetch<0>(A.Rp);
            for (int k = 0; k < K; ++k)
                eval_cluster(k, (k + 1 < K) ? A.Rp + (size_t)(3 * (k + 1)) * C::MATSZ : nullptr);
            }
        } else {
            // Screened label phase, workgroup-wide (the fragment staging is shared by the four waves):
            //  (1) reference clusters k0 / k1 = previous labels of the workgroup's first and last point, evaluated in full (k1 only
            //      when it differs: a tile that straddles two clusters of the label-sorted order -- with ONE reference the points
            //      of the other cluster had a hopeless threshold, nothing could be excluded for them and all K clusters were
            //      evaluated: 31 such tiles per sweep at K = 32, 1.6 M cycles each against 0.18 M for an ordinary tile);
            //  (2) every wave tail-screens all other clusters against its own points; a cluster is evaluated if ANY wave
            //      could not exclude it (bit in survm); (3) survivors are evaluated in index order.
            __syncthreads();                                   // survm / sh_first / sh_last of the previous tile are no longer read
            if (tid < DPMM_MAX_CLUSTERS_K / 32) { survm[tid] = 0u; surv2[tid] = 0u; }
            {
                int prev = (valid && A.use_prev) ? ((my_prev_bin != -2 ? my_prev_bin : A.bins[myp]) >> 1) : -1;
                if ((unsigned)prev >= (unsigned)K) prev = -1;
                const unsigned long long pm = __ballot(prev >= 0);
                int kf = -1, kl = -1;
                if (pm) { kf = __shfl(prev, __ffsll((long long)pm) - 1); kl = __shfl(prev, 63 - __clzll((long long)pm)); }
                if (lane == 0) { sh_first[wave] = kf; sh_last[wave] = kl; }
            }
            __syncthreads();
            k0 = 0;
            for (int wv = 3; wv >= 0; --wv) if (sh_first[wv] >= 0) k0 = sh_first[wv];
            k1 = k0;
            for (int wv = 0; wv < 4; ++wv) if (sh_last[wv] >= 0) k1 = sh_last[wv];
            const int nref = k1 != k0 ? 2 : 1;
            float aref = -INFINITY;
            // Reference bracket (niw_bracket_big_kernel, a launch of its own in front of this one): every point of the WORKGROUP was in k0 -> aref =
            // the lower end of a certified bracket of a_k0, nothing is recorded; k0's Float32 evaluation follows behind the screens only if some
            // cluster survives them for some wave (else the draw returns k0 whatever the exact value is: the one-cluster draw below).
            bool bracketed = false, ref_skipped = false;
            if constexpr (NB >= 8) {
                // (launch_niw_bracket_big ran in front of this launch: A.sp_frag = one word per tile -- 1 + k0 where every point of the tile
                // was in k0, else 0 --, A.sp_cons = the bracket's lower end of a_k0 per position of the visiting order)
                if (A.sp_frag != nullptr && nref == 1 && A.sp_frag[tile] == (uint32_t)(k0 + 1)) {
                    bracketed = true;
                    aref = valid ? A.sp_cons[mypos] : -INFINITY;
                    ++nw_br;
                }
            }
            if (!bracketed) {
            ev.template prefetch<0>(A.Rp + (size_t)(3 * k0) * C::MATSZ);
            for (int rr = 0; rr < nref; ++rr) {       // one call site: the unrolled evaluation exists once for both references
                const float a = eval_cluster(rr ? k1 : k0, (rr + 1 < nref) ? A.Rp + (size_t)(3 * k1) * C::MATSZ : nullptr);
                if (a > aref) aref = a;                // a NaN never raises the reference
            }
            }
            STAMP(s2);
            const float my_thr = valid ? aref - A.screen_margin : INFINITY;
            f32x4 xt = (f32x4){0.f, 0.f, 0.f, 0.f};
            {
                const int src = ci + 16 * A.tail_g;
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    f32x4 v;
                    v.x = __shfl(x[n][NB - 1].x, src); v.y = __shfl(x[n][NB - 1].y, src);
                    v.z = __shfl(x[n][NB - 1].z, src); v.w = __shfl(x[n][NB - 1].w, src);
                    if (g == n) xt = v;                        // owner lane 16 n + ci holds point (n, ci)
                }
            }
            // cluster-per-lane ball test first (see ball_far: the wave's points lie in a ball around the tail mean of the first reference
            // cluster), then the per-point pair screen for the clusters it leaves -- as in the D <= 64 kernel
            BallWave ball; ball.ok = false;
            if (A.ball) ball = ball_of_wave(A.tail, K, k0, xt, my_thr, valid);
            for (int base = 0; base < K; base += 64) {
                unsigned long long cand = (K - base >= 64) ? ~0ull : ((1ull << (K - base)) - 1ull);
                if (k0 >= base && k0 < base + 64) cand &= ~(1ull << (k0 - base));
                if (k1 >= base && k1 < base + 64) cand &= ~(1ull << (k1 - base));
                if (ball.ok) cand &= ~ball_far(A.tail, K, base, lane, ball);
                for (unsigned long long pend = cand; pend;) {
                    const int sh = __builtin_ctzll(pend) & ~1;
                    pend &= ~(3ull << sh);
                    ++nw_tail;
                    cand &= ~((unsigned long long)tail_pair_far(tail_load_pair(A.tail, (base + sh) >> 1), xt, my_thr) << sh);
                }
                if (lane == 0) {
                    const uint32_t lo = (uint32_t)cand, hi = (uint32_t)(cand >> 32);
                    if (lo) atomicOr(&survm[base >> 5], lo);
                    if (hi) atomicOr(&survm[(base >> 5) + 1], hi);
                }
            }
            __syncthreads();
            // (2b) the clusters the tail screen left over get the 16-row screen of the D <= 64 kernel before anybody evaluates them in
            // full: the LAST 16 rows of y = R z need the last 16 features only (4 NG matrix instructions, the block's fragment straight
            // from L2), the sum over a lane's 4 rows is a lower bound of q.  Every wave tests its own points; a cluster stays when some
            // wave cannot exclude it (surv2).  Before: 0.43 left-over clusters per tile, each a full evaluation of 1088 matrix
            // instructions -- 11 % of the D = 256 launch.
            {
                float thr_n[NG];
#pragma unroll
                for (int n = 0; n < NG; ++n) thr_n[n] = __shfl(my_thr, 16 * n + ci);
                const int nws = (K + 31) >> 5;
                for (int w2 = 0; w2 < nws; ++w2) {
                    uint32_t sb = __builtin_amdgcn_readfirstlane(survm[w2]);
                    for (; sb; sb &= sb - 1u) {
                        const int k = (w2 << 5) + __builtin_ctz(sb);
                        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(A.Rp + (size_t)(3 * k) * C::MATSZ + (size_t)(C::NP - 1) * 256 + lane * 4);
                        const f32x4 m4 = *reinterpret_cast<const f32x4 *>(A.mup + (size_t)(3 * k) * C::DP + 16 * (NB - 1) + 4 * g);
                        const float ck = A.cst[3 * k];
                        bool skip = true;
#pragma unroll
                        for (int n = 0; n < NG; ++n) {
                            const f32x4 zz = x[n][NB - 1] - m4;
                            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], zz[jj], acc, 0, 0, 0);
                            float ql = acc[0] * acc[0];
                            ql = __builtin_fmaf(acc[1], acc[1], ql); ql = __builtin_fmaf(acc[2], acc[2], ql); ql = __builtin_fmaf(acc[3], acc[3], ql);
                            unsigned long long mk = __ballot(__builtin_fmaf(-0.5f, ql, ck) < thr_n[n]);      // a point is covered if ANY of its 4 row-group lanes proves the bound
                            mk |= mk >> 32;
                            mk |= mk >> 16;
                            skip = skip && ((mk & 0xFFFFull) == 0xFFFFull);
                        }
                        ++nw_scr;
                        if (!skip && lane == 0) atomicOr(&surv2[k >> 5], 1u << (k & 31));
                    }
                }
            }
            __syncthreads();
            if (tid < DPMM_MAX_CLUSTERS_K / 32) survm[tid] = surv2[tid];
            __syncthreads();
            STAMP(s3);
#ifdef DPMM_STAMPS
            T_ref += s2 - s1; T_tail += s3 - s2; sv0 = s3;
#endif
            const int nw = (K + 31) >> 5;
            int w = 0;
            uint32_t bits = __builtin_amdgcn_readfirstlane(survm[0]);
            auto next_surv = [&]() -> int {
                while (bits == 0u) {
                    ++w;
                    if (w >= nw) return -1;
                    bits = __builtin_amdgcn_readfirstlane(survm[w]);
                }
                const int b = __builtin_ctz(bits);
                bits &= bits - 1u;
                return (w << 5) + b;
            };
            int kc = next_surv(), kpend = -1;
            if (bracketed) {
                if (kc >= 0) { kpend = kc; kc = k0; }      // somebody survived: the exact value of k0 (table entry of the draw) first, through the same call site
                else {
                    ref_skipped = true;
                    m_run = 0.f; best = k0;          // (a finite stand-in: the one-cluster draw never reads the table)
                }
            }
            if (kc >= 0) ev.template prefetch<0>(A.Rp + (size_t)(3 * kc) * C::MATSZ);
            while (kc >= 0) {
                int kn;
                if (kpend >= 0) { kn = kpend; kpend = -1; } else kn = next_surv();
                eval_cluster(kc, kn >= 0 ? A.Rp + (size_t)(3 * kn) * C::MATSZ : nullptr);
                kc = kn;
            }
            ref_skipped_tile = ref_skipped;
        }

        // ---- label draw (owner lanes), src/utils.jl:19-31
        STAMP(s4);
        int z = 0;
        float u_sub = 0.f;
        if (valid) {
            const Philox4 r = philox4x32_10(A.seed, (uint64_t)(A.first_index + myp), A.epoch, STREAM_SWEEP);
            u_sub = u01(r.v[1]);
            if (A.final_argmax) {
                z = best;
            } else if (ref_skipped_tile) {
                // one evaluated cluster, k0: s = exp_det(0) = 1, t = u, the scan stops at k0 -- or at index 0 when t <= 0, as the full scan does
                z = (u01(r.v[0]) * 1.0f <= 0.f) ? 0 : k0;
            } else if (m_run == -INFINITY) {
                z = 0;
            } else if (screening) {
                // skipped clusters contribute exact zeros: visit the evaluated ones only, in index order (bit-identical
                // to the full scan over a table holding -inf for them)
                const int nw = (K + 31) >> 5;
                float s = 0.f;
                for (int w = 0; w < nw; ++w) {
                    uint32_t bb = survm[w] | ((k0 >> 5) == w ? (1u << (k0 & 31)) : 0u) | ((k1 >> 5) == w ? (1u << (k1 & 31)) : 0u);
                    for (; bb; bb &= bb - 1u) s += exp_det(nan_to_ninf(scr[(int64_t)((w << 5) + __builtin_ctz(bb)) * sstride]) - m_run);
                }
                const float t = u01(r.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                bool found = false;
                for (int w = 0; w < nw && !found; ++w) {
                    uint32_t bb = survm[w] | ((k0 >> 5) == w ? (1u << (k0 & 31)) : 0u) | ((k1 >> 5) == w ? (1u << (k1 & 31)) : 0u);
                    for (; bb; bb &= bb - 1u) {
                        const int k = (w << 5) + __builtin_ctz(bb);
                        cw += exp_det(nan_to_ninf(scr[(int64_t)k * sstride]) - m_run);
                        if (!(cw < t)) { z = k; found = true; break; }
                    }
                }
                if (t <= 0.f) z = 0;      // the full scan stops at k = 0 when the threshold is already met there
            } else {
                float s = 0.f;
                for (int k = 0; k < K; ++k) s += exp_det(nan_to_ninf(scr[(int64_t)k * sstride]) - m_run);
                const float t = u01(r.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                for (int k = 0; k < K; ++k) {
                    cw += exp_det(nan_to_ninf(scr[(int64_t)k * sstride]) - m_run);
                    if (!(cw < t)) { z = k; break; }
                }
            }
        }
        if (A.labels_only) continue;  // debug_loglik: table only (uniform branch)

        // ---- phase 2: sub-labels.  Distinct labels of the workgroup -> LDS bitmap
        if (queue && tid == 0) sh_next = pend;               // the next tile (claimed at the top of this one: arrived long ago)
        __syncthreads();
        if (queue) {      // the next tile's visiting order and previous labels, consumed at the top of the next trip
            nx_tile = sh_next;
            if (nx_tile < A.ntiles) {
                const int64_t nb = nx_tile * C::TILE + (int64_t)wave * C::WPTS;
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    const int64_t pos = nb + 16 * n + ci;
                    pf_ord[n] = (pos < A.n && use_order) ? A.order[pos] : 0;
                }
                const int64_t mp = nb + lane;
                const bool nvalid = owner && mp < A.n;
                pf_my = (nvalid && use_order) ? A.order[mp] : 0;
                pf_bin = nvalid ? -2 : -1;                   // (-2: previous label to be fetched at the end of this trip, when pf_my has arrived)
            }
        }
        STAMP(s5);
        if (tid < DPMM_MAX_CLUSTERS_K / 32) present[tid] = 0u;
        __syncthreads();
        if (valid) atomicOr(&present[z >> 5], 1u << (z & 31));
        __syncthreads();
        float b0 = -INFINITY, b1 = -INFINITY;
        const int nwords = (K + 31) >> 5;
        // find the first label, then walk the set with one-label lookahead for the prefetch
        int w = 0;
        uint32_t bits = __builtin_amdgcn_readfirstlane(present[0]);
        auto next_label = [&](int &ww, uint32_t &bb) -> int {
            while (bb == 0u) {
                ++ww;
                if (ww >= nwords) return -1;
                bb = __builtin_amdgcn_readfirstlane(present[ww]);
            }
            const int b = __builtin_ctz(bb);
            bb &= bb - 1u;
            return (ww << 5) + b;
        };
        int kcur = next_label(w, bits);
        if (kcur >= 0) ev.template prefetch<0>(A.Rp + (size_t)(3 * kcur + 1) * C::MATSZ);
        while (kcur >= 0) {
            const int knext = next_label(w, bits);
            const bool wave_has = __any(valid && z == kcur);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int j = 3 * kcur + 1 + s;
                const float *Rcur = A.Rp + (size_t)j * C::MATSZ;
                const float *Rnext = (s == 0) ? A.Rp + (size_t)(j + 1) * C::MATSZ
                                              : (knext >= 0 ? A.Rp + (size_t)(3 * knext + 1) * C::MATSZ : nullptr);
                f32x4 mu[NB];
#pragma unroll
                for (int t = 0; t < NB; ++t)
                    mu[t] = *reinterpret_cast<const f32x4 *>(A.mup + (size_t)j * C::DP + 16 * t + 4 * g);
                float q[NG];
#pragma unroll
                for (int n = 0; n < NG; ++n) q[n] = 0.f;
                eval_matrix<NB, NG, CH>(ev, lds, Rcur, Rnext, x, mu, q, lane, wave_has);
                if (wave_has) ++nw_full;
                const float qs = reduce_select<NG>(q, g);
                const float b = __builtin_fmaf(-0.5f, qs, A.cst[j]);
                if (valid && z == kcur) {
                    if (s == 0) b0 = b; else b1 = b;
                }
            }
            kcur = knext;
        }
        if (valid) {
            const int sl = draw2(b0, b1, u_sub);
            A.bins[myp] = 2 * z + sl;
        }
        if (queue && nx_tile >= 0 && nx_tile < A.ntiles && pf_bin == -2) {
            // the previous label of the owner's point of the NEXT tile (final: this tile's draws touch this tile's points only); the
            // load is in flight across the tile boundary and consumed after the next X gather
            const int64_t mp = nx_tile * C::TILE + (int64_t)wave * C::WPTS + lane;
            pf_bin = A.use_prev ? A.bins[use_order ? (int64_t)pf_my : mp] : -1;
        }
        STAMP(s6);
#ifdef DPMM_STAMPS
        T_x += s1 - s0; T_surv += s4 - sv0; T_draw += s5 - s4; T_p2 += s6 - s5; T_tot += s6 - s0; ++ntile;
        if (s6 - s0 > T_max) T_max = s6 - s0;
        if (!T_first) T_first = s0;
        T_last = s6;
#endif
    }
    // executed-work counters: one 32-byte slot per wave, plain read-modify-write by its owner (they add up over launches until
    // dpmm_last_sweep_work reads and clears them: a benchmark reads once after its timed loop instead of once per step).  (They were four atomic adds per wave on one cache line: waves
    // finish in bursts -- all those with one tile fewer than the rest at the same moment -- and a burst of same-address device-scope
    // atomics held up the loads of the waves still running: the last round of the D <= 64 kernel took 4x as long, a constant ~70 us
    // per launch whatever N, 20 % of the launch at the 8-GPU shard size.)
    if (A.work && lane == 0) {
        unsigned long long *slot = A.work + DPMM_WORK_SLOTS + ((size_t)blockIdx.x * 4 + wave) * DPMM_WORK_PER_WAVE;      // (accumulates over launches; cleared by the reader)
        slot[0] += nw_tiles; slot[1] += nw_full; slot[2] += nw_scr; slot[3] += nw_tail; slot[4] += nw_br;
    }
#ifdef DPMM_STAMPS
    if (A.dbg && lane == 0 && blockIdx.x < 4096) {
        unsigned long long *d = A.dbg + ((size_t)blockIdx.x * 4 + wave) * 16;
        d[0] = T_x; d[1] = T_ref; d[2] = 0; d[3] = T_tail; d[4] = T_surv; d[5] = T_draw; d[6] = T_p2; d[7] = T_tot;
        d[8] = ntile; d[9] = nw_tail; d[10] = T_max; d[11] = T_first; d[12] = T_last; d[13] = 0; d[14] = 0;
    }
#endif
}


// ---------------------------------------------------------------------------------------
// "Direct" variant for D <= 64 (NB <= 4): no LDS, no barriers.  A packed factor is only
// NP KiB (10 KiB at D=64) and is re-read by every wave from L2 (~4 B/clk/CU, far below the
// L2 rate), so each wave streams its own A fragments global -> registers one matrix ahead of
// the MFMAs and never synchronises with its neighbours: waves drift apart and fill each
// other's epilogue / draw phases on the shared matrix pipe, and the sub-label phase walks the
// distinct labels of the wave's own 64 points (not of the whole workgroup).
// Streaming evaluation of one packed matrix: row-block bi is computed from `cur` while the
// fragments of the next row-block (or of row-block 0 of the next matrix, plus its mu) are in
// flight.  On entry rb0/mu hold row-block 0 and mu of THIS matrix; on exit those of Rnext.
#define DPMM_PRIO_ARG prio
// EARLY (survivors of the screens only): after the FIRST row block -- rows 0..15 of y = R z, the rows that see every feature and, for an
// ill-conditioned Sigma, carry most of the quadratic form (R_ii^2 is the precision of feature i GIVEN the features behind it: largest at
// the top; the last rows, which the cheap screens use, are marginal precisions) -- the partial sum is a lower bound of q: if
// cst - q_partial / 2 is below thr[n] for every point, the cluster is excluded after 64 + 4 matrix instructions instead of 164 + 4 and
// *excluded is set (rb0 / mu then still hold THIS matrix's first row block: the caller reloads).  The accumulation chain of q is the one
// of the plain evaluation, so a cluster that is not excluded gets bit-identical values.
template <int NB, int NG, bool EARLY = false>
__device__ __forceinline__ float quad_stream(const float *__restrict__ Rm, const float *__restrict__ Rnext,
                                             const float *__restrict__ mup_next, f32x4 (&rb0)[NB], f32x4 (&mu)[NB],
                                             const f32x4 (&x)[NG][NB], int lane, int g, bool active, float (&tot_all)[NG], int prio = 0,
                                             float cst = 0.f, const float *thr = nullptr, bool *excluded = nullptr) {
    float q[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) q[n] = 0.f;
    if (DPMM_PRIO_ARG) __builtin_amdgcn_s_setprio(0);
    // Fragment schedule (everything is unrolled; F[] are SSA values, nothing is copied).  The MFMA time of a row-block
    // shrinks with bi (NB - bi pairs) while the L2 latency of its fragments does not, so the later row-blocks are fetched
    // earlier than "one ahead": rb1 during rb0, ALL remaining row-blocks during rb1, and row-block 0 / mu of the next
    // matrix during the second-to-last row-block.
    constexpr int NPF = NB * (NB + 1) / 2;
    f32x4 F[NPF], nx0[NB], mun[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) { F[t] = rb0[t]; nx0[t] = rb0[t]; mun[t] = mu[t]; }
    auto load_rowblock = [&](int b) {
#pragma unroll
        for (int t = 0; t < NB - b; ++t)
            F[pair_base<NB>(b) + t] = *reinterpret_cast<const f32x4 *>(Rm + (pair_base<NB>(b) + t) * 256 + lane * 4);
    };
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        if (bi == 0 && NB > 1) load_rowblock(1);
        if (bi == 1) {
#pragma unroll
            for (int b = 2; b < NB; ++b) load_rowblock(b);
        }
        if (bi == (NB >= 2 ? NB - 2 : 0) && Rnext) {
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                nx0[t] = *reinterpret_cast<const f32x4 *>(Rnext + t * 256 + lane * 4);
                mun[t] = *reinterpret_cast<const f32x4 *>(mup_next + 16 * t + 4 * g);
            }
        }
        if (active) {
            f32x4 acc[NG];
#pragma unroll
            for (int n = 0; n < NG; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = bi; t < NB; ++t) {
                const f32x4 a = F[pair_base<NB>(bi) + (t - bi)];
                f32x4 zz[NG];
#pragma unroll
                for (int n = 0; n < NG; ++n) zz[n] = x[n][t] - mu[t];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                    for (int n = 0; n < NG; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], zz[n][jj], acc[n], 0, 0, 0);
                }
            }
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                q[n] = __builtin_fmaf(acc[n][0], acc[n][0], q[n]);
                q[n] = __builtin_fmaf(acc[n][1], acc[n][1], q[n]);
                q[n] = __builtin_fmaf(acc[n][2], acc[n][2], q[n]);
                q[n] = __builtin_fmaf(acc[n][3], acc[n][3], q[n]);
            }
        }
        if constexpr (EARLY && NB > 1) {
            if (bi == 0) {
                bool out = true;
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    const f32x4 tot = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, q[n], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    out = out && (__builtin_fmaf(-0.5f, tot[0], cst) < thr[n]);       // thr[n] = +inf for columns without a point
                }
                if (__all(out)) {
                    *excluded = true;
                    if (DPMM_PRIO_ARG) __builtin_amdgcn_s_setprio(2);
                    return 0.f;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) { rb0[t] = nx0[t]; mu[t] = mun[t]; }
    // Cross-lane sum over the 4 row groups g without leaving the matrix pipe: with A = ones,
    // D[i][c] = sum_g B[g][c], i.e. every lane of column c receives the column total.
    float sel = 0.f;
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        const f32x4 tot = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, q[n], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        tot_all[n] = tot[0];       // quadratic form of point (n, column of this lane): same value in all 4 row-group lanes
        if (g == n) sel = tot[0];
    }
    if (DPMM_PRIO_ARG) __builtin_amdgcn_s_setprio(2);
    return sel;
}

template <int NB>
__device__ __forceinline__ void load_rb0(const float *__restrict__ Rm, const float *__restrict__ mup, f32x4 (&rb0)[NB],
                                         f32x4 (&mu)[NB], int lane, int g) {
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        rb0[t] = *reinterpret_cast<const f32x4 *>(Rm + t * 256 + lane * 4);
        mu[t] = *reinterpret_cast<const f32x4 *>(mup + 16 * t + 4 * g);
    }
}

// FAST: the steady-state configuration known at launch time (screening with the tail screen, no far mask, LDS table, no
// Student-t mode, no table output): the mode tests below become compile-time constants -- fewer live scalars (the generic
// kernel spills > 100 SGPRs into VGPR lanes) and fewer branches per tile.  Same arithmetic, same results.
// DIR: the instantiation with the direction screen (direction_far) in it -- a kernel of its own because its registers (the tile's z0 as bf16
// for all four point groups) would cost the common kernel 40 spilled registers and 15 % of its time for a branch it never takes.
// LSTORE (D in 33 .. 64 while the bf16 sub-label evaluation of niw_lean.hip is active): the instantiation stops behind the label draw and
// STORES the labels (bins = 2 z; the sub-label bit is a placeholder) -- niw_sub_kernel draws the sub-labels of the same tiles in a launch of
// its own.  The tiles come from a list (A.tdf re-used as `const uint32_t *`: [0] = count, [1 ..] = wave-tile indices; Student-t mode does
// not exist in these instantiations) or, with a null list, from the usual schedule.
// LIST (with LSTORE): A.tdf names a list of spans to process instead of every tile -- an instantiation of its own, so that the all-tiles
// LSTORE kernels do not carry the list's state (15 instead of 4 spilled registers, all-tiles sweep 1.53 -> 1.64 ms when they did)
template <int NB, int NG, int OCC, bool FAST = false, bool DIR = false, bool LSTORE = false, bool LIST = false>
__global__ __launch_bounds__(256, OCC) void niw_sweep_direct_kernel(NiwSweepArgs A) {
    const float *const a_tdf = LSTORE ? nullptr : A.tdf;
    static_assert(!LIST || LSTORE, "a list belongs to the label-storing instantiations");
    const uint32_t *const tlist = LIST ? reinterpret_cast<const uint32_t *>(A.tdf) : nullptr;
    if constexpr (LIST) {
        // most sweeps hand on nothing (and a workgroup whose four waves are beyond the list has nothing to do either): leave before the staging below;
        // the list's length still goes to the host (A.mdist: see further down)
        const uint32_t lc = tlist[0];
        if (lc == 0u || blockIdx.x * 4u >= lc) {
            if (A.mdist && blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<uint32_t *>(const_cast<float *>(A.mdist)) = lc;
            return;
        }
    }
#ifdef DPMM_STAMPS
    unsigned long long T_x = 0, T_quad = 0, T_epi = 0, T_draw = 0, T_p2 = 0, T_tot = 0, N_scr = 0, N_tail = 0, T_prep = 0, T_far = 0, T_surv = 0, T_init = 0, T_i1 = 0, T_i2 = 0, T_lastd = 0, T_long = 0, T_longat = 0, T_firstd = 0; int ntile = 0;
#endif
    constexpr int DP = 16 * NB, NP = NB * (NB + 1) / 2, MATSZ = NP * 256, WPTS = 16 * NG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int ci = lane & 15, g = lane >> 4;
    const int K = A.K;
    if (A.prio) __builtin_amdgcn_s_setprio(2);
    const bool use_order = A.order != nullptr && *A.order_total == (int32_t)A.n;
    const bool owner = lane < WPTS;
    const int nwtiles = (int)((A.n + WPTS - 1) / WPTS);          // one tile per wave (32-bit tile indices: 2^31 tiles of 64 points)
    const int wave_id = (int)blockIdx.x * 4 + (tid >> 6);
    const int nwaves = (int)gridDim.x * 4;
    __shared__ uint32_t surv_bits[4][32];   // per wave: clusters that survived the screen (K <= 1024)
    __shared__ uint32_t eval_bits[4][32];   // per wave: reference clusters + survivors
    extern __shared__ __attribute__((aligned(16))) float lds_tab[];
    // wave-private [K][WPTS] table of a_k in LDS when it fits (A.lds_rows >= K), else the global scratch
    const bool tab_lds = FAST || (A.lds_rows >= K && !A.scratch_by_tile);
    float *ltab = lds_tab + (size_t)(tid >> 6) * A.lds_rows * WPTS + lane;
    // K beyond the LDS budget (lds_rows < K): the rows go to the global scratch as before, and a COMPACT copy of the evaluated clusters'
    // rows (lds_rows slots per wave, slot_of[k] names a cluster's slot) serves the draw of a screened tile -- which visits evaluated
    // clusters only.  (Without it a K = 256 sweep spent 3 K dependent global accesses per point on -inf rows: 12.7 ms at N = 1e7.)
    const bool compact = !FAST && !tab_lds && A.lds_rows > 0 && !A.scratch_by_tile;
    uint8_t *slot_of = reinterpret_cast<uint8_t *>(lds_tab + (size_t)4 * A.lds_rows * WPTS + (A.screen_lds ? (size_t)K * (256 + 16) : 0)) +
                       (size_t)(tid >> 6) * ((K + 3) & ~3);
    // screen operands of all K clusters (last fragment pair 1 KiB + last 16 means), staged once per workgroup
    float *scrA = lds_tab + (size_t)4 * A.lds_rows * WPTS;
    float *scrM = scrA + (size_t)K * 256;
    if (A.screen_lds) {
        for (int e = tid; e < K * 64; e += 256)
            reinterpret_cast<f32x4 *>(scrA)[e] = *reinterpret_cast<const f32x4 *>(A.Rp + ((size_t)(3 * (e >> 6)) * NP + (NP - 1)) * 256 + (e & 63) * 4);
        for (int e = tid; e < K * 4; e += 256)
            reinterpret_cast<f32x4 *>(scrM)[e] = *reinterpret_cast<const f32x4 *>(A.mup + (size_t)(3 * (e >> 2)) * DP + 16 * (NB - 1) + 4 * (e & 3));
        __syncthreads();
    }

    // Tile schedule: STATIC rounds (tile = wave + r * waves) for most of the launch, then a QUEUE for the last rounds (A.work[4], cleared
    // with the work counters; non-table sweeps only).  Why: at N = 1e7 every wave but one ends within 3.5 % of the median, and that one
    // runs 8 % long -- a single tile of the 156 250 in which one point has a hopeless reference costs 30 full evaluations (8 tile times)
    // and, statically assigned, delays the end of the launch by all of it (per-wave longest-tile stamps, scripts/stamps_bench.py).  With
    // the last max(2, rounds / 8) rounds handed out tile by tile the other waves absorb it (launches of fewer than 4 rounds stay static:
    // there the claims cost more than the balance gains; DPMM_OPT_SWEEP_QUEUE_ROUNDS).  A claim is an atomic add with return, issued
    // one tile AHEAD of its use (the value is read when the tile after next is set up), so its latency is never waited for; it sits in
    // the in-order VMEM queue in front of the tile's X gather, which takes as long.  Results do not depend on who takes a tile (the
    // random streams are keyed by the point).  History: claiming EVERY tile (156 k atomics per launch) cost 14 %; that measurement,
    // like the first 15/16-static attempt (no gain), was taken while the end-of-wave counter atomics still stalled the last rounds.
    // (A start stagger of the odd hardware wave slots was measured twice: no effect.)
    // Cross-tile prefetch (static tile schedule): while a tile is processed, the wave already fetches the NEXT
    // tile's point indices (order[]) and their previous labels (bins[]); the next tile then issues its X gather at
    // once instead of walking the dependent chain order -> X, order -> bins -> reference cluster -> fragments.
    // (Also touching the next tile's X lines to pull them into L2 was measured: no gain, +30 % HBM traffic.)
    int nx_p = -1, nx_bin = -1;
    int nx_tile = -1;
    unsigned nw_tiles = 0, nw_full = 0, nw_scr = 0, nw_tail = 0, nw_br = 0, nw_bb = 0, nw_bt = 0;   // executed-work counters of this wave (wave-uniform); nw_bb / nw_bt: bf16 bottom / top screens
    unsigned nw_b3l = 0;                      // LIST: three-plane sub-cluster evaluations (the high half of work slot 7, as niw_lean.hip's kernels count them)
    unsigned nw_sp = 0, nw_cand = 0;          // direction screens run (DIR); candidates behind the 4-row tests (DIR: counted; else = nw_bb, every one gets a bottom screen)
    unsigned nw_dcand = 0, nw_dexcl = 0;      // DIR: candidates the direction screens were given, and how many of them they removed (the screen's yield)
    const int rounds_all = nwtiles / nwaves;
    int dyn_rounds = A.queue_rounds >= 0 ? A.queue_rounds : (rounds_all < 4 ? 0 : (rounds_all / 8 > 2 ? rounds_all / 8 : 2));
    if (dyn_rounds > rounds_all) dyn_rounds = rounds_all;
    const bool dyn_ok = A.work != nullptr && dyn_rounds > 0;                          // (dyn_rounds == 0: the static schedule, also for the last partial round)
    const int dyn0 = dyn_ok ? (rounds_all - dyn_rounds) * nwaves : nwtiles;      // tiles >= dyn0 come from the queue
    // Eight queue heads, one per 128-byte line (same-address atomics are served at ~250 M/s: 2048 waves claiming a tile each per round
    // through ONE counter cost 8 us per round, a third of a round's work -- measured at the 8-GPU shard size).  Queue q hands out the
    // tiles dyn0 + 8 c + q; a wave starts at queue (wave & 7) and moves on to the next one when its queue is exhausted.
    unsigned long long q_pend = 0;            // lane 0: the claim in flight (count c of queue q_cur)
    bool q_inflight = false;
    int q_cur = (int)(wave_id & (DPMM_WORK_QUEUES - 1)), q_dead = 0;      // current queue; queues found exhausted in a row
    auto q_issue = [&]() {
        if (lane == 0) q_pend = atomicAdd(&A.work[8 + 16 * q_cur], 1ull);
        q_inflight = true;
    };
    auto q_take = [&]() -> int {         // waits for the claim issued one tile ago; an exhausted queue is replaced by the next one
        for (;;) {
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q_pend), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q_pend >> 32));
            q_inflight = false;
            const long long t = (long long)dyn0 + (long long)(((unsigned long long)hi << 32) | lo) * DPMM_WORK_QUEUES + q_cur;
            if (t < nwtiles) { q_dead = 0; return (int)t; }
            if (++q_dead >= DPMM_WORK_QUEUES) return -1;
            q_cur = (q_cur + 1) & (DPMM_WORK_QUEUES - 1);
            q_issue();                        // (waited for at once: only at the very end of the launch)
        }
    };
    // reference bracket: after two waves' worth of tiles in a row on which it decided nothing (overlapping clusters: something always survives
    // the screens) the wave goes straight to the Float32 evaluation for the next eight tiles, then tries again
    int br_fail = 0, br_skip = 0;
    int tile0, tnext_v = -1;
    // LSTORE with a list: entry i = (first position in the visiting order, number of positions <= 64) at tlist[1 + 2 i], tlist[2 + 2 i] -- the
    // lean kernel's tiles are aligned to the bins of the sort, not to multiples of 64.  `tile` is then the entry's index (static stride over the
    // list), l_pos / l_end the current entry's span; every validity test below compares with n_end (= A.n without a list).
    int li = wave_id;
    const int lcount = LIST ? (int)tlist[0] : 0;
    const int tile_sent = LIST ? 0x7fffffff : nwtiles;      // "no further tile"
    // (the list's length for the host's regime decision: A.mdist -- the pre-screen's table, never read when a list is walked (lam == nullptr) --
    // carries a pinned word for it; a field of its own would move every instantiation's scalars)
    if constexpr (LIST) { if (A.mdist && blockIdx.x == 0 && tid == 0) *reinterpret_cast<uint32_t *>(const_cast<float *>(A.mdist)) = tlist[0]; }
    int l_pos = 0, l_end = 0;                  // (wave-uniform, positions < 2^31)
    if (LIST) {
        tile0 = li < lcount ? li : -1;
        if (tile0 >= 0) { l_pos = __builtin_amdgcn_readfirstlane((int)tlist[1 + 2 * li]); l_end = l_pos + __builtin_amdgcn_readfirstlane((int)tlist[2 + 2 * li]); }
    }
    else if (!dyn_ok) tile0 = wave_id < nwtiles ? wave_id : -1;
    else if (wave_id < dyn0) tile0 = wave_id;
    else { q_issue(); tile0 = q_take(); if (tile0 >= 0) q_issue(); }
    for (int tile = tile0; tile >= 0; tile = tnext_v) {
        const int64_t wbase = LIST ? (int64_t)l_pos : (int64_t)tile * WPTS;
        const int64_t n_end = LIST ? (int64_t)l_end : A.n;
        ++nw_tiles;
        STAMP(s0);
        const int64_t mypos = wbase + lane;    // position in processing order
        const bool valid = owner && mypos < n_end;
        const bool prefetched = nx_tile == tile;
        const bool screening = FAST || (NB >= 2 && A.screen_margin > 0.f && !a_tdf && !A.scratch_by_tile && K > 1);
        int myp32, binv = -1;
        if (prefetched) {
            myp32 = nx_p; binv = nx_bin;
        } else {      // first tile of the wave (or dynamic tile schedule): the dependent chain, before anything else is in flight
            myp32 = valid ? (use_order ? A.order[mypos] : (int)mypos) : -1;
            if (screening && myp32 >= 0 && A.use_prev) binv = A.bins[myp32];
        }
        const int64_t myp = myp32 >= 0 ? (int64_t)myp32 : mypos;   // the point this lane draws for
        // reference clusters of the screened label phase: previous labels of the wave's first and last point
        int k0 = 0, k1 = 0;
        if (screening) {
            int prev = binv >= 0 ? (binv >> 1) : -1;
            if ((unsigned)prev >= (unsigned)K) prev = -1;
            const unsigned long long pm = __ballot(prev >= 0);
            if (pm) {
                k0 = __shfl(prev, __ffsll((long long)pm) - 1);
                k1 = __shfl(prev, 63 - __clzll((long long)pm));
            }
            k0 = __builtin_amdgcn_readfirstlane(k0);
            k1 = __builtin_amdgcn_readfirstlane(k1);
        }
        f32x4 x[NG][NB];
#pragma unroll
        for (int n = 0; n < NG; ++n) {
#ifdef DPMM_EXP_XCACHE       // experiment: every tile gathers from the first 4096 points (cache-resident X): what does the HBM gather cost?
            const int pn0 = __shfl(myp32, 16 * n + ci);
            const int pn = pn0 >= 0 ? (pn0 & 4095) : pn0;
#else
            const int pn = __shfl(myp32, 16 * n + ci);
#endif
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                const int e = 16 * t + 4 * g;
                x[n][t] = (pn >= 0 && e < A.ldx) ? *reinterpret_cast<const f32x4 *>(A.X + (int64_t)pn * A.ldx + e)
                                                  : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        int tnext;
        int nl_pos = 0, nl_end = 0;
        if (LIST) {
            li += nwaves; tnext = li < lcount ? li : tile_sent;
            if (li < lcount) { nl_pos = __builtin_amdgcn_readfirstlane((int)tlist[1 + 2 * li]); nl_end = nl_pos + __builtin_amdgcn_readfirstlane((int)tlist[2 + 2 * li]); }
        }
        else if (!dyn_ok) tnext = tile + nwaves < nwtiles ? tile + nwaves : nwtiles;
        else if (tile + nwaves < dyn0) tnext = tile + nwaves;                        // static successor
        else {
            if (!q_inflight) q_issue();                                               // first tile from the queue: this one claim is waited for
            tnext = q_take();
            if (tnext >= 0) q_issue();                                                // the tile after next, read one tile from now
            else tnext = nwtiles;
        }
        tnext_v = tnext < tile_sent ? tnext : -1;
        int pf_p = -1, pf_bin = -1;
        if (tnext < tile_sent) {
            const int64_t posn = (LIST ? (int64_t)nl_pos : (int64_t)tnext * WPTS) + lane;
            if (owner && posn < (LIST ? (int64_t)nl_end : A.n)) pf_p = use_order ? A.order[posn] : (int)posn;
        }
#ifdef DPMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(s1);
        float *scr = FAST ? nullptr : A.scratch + (A.scratch_by_tile ? (int64_t)tile * WPTS : (int64_t)wave_id * WPTS) + lane;
        const int64_t sstride = A.scratch_stride;

        float m_run = -INFINITY;
        int best = 0;
        bool nan_seen = false;
        bool ref_skipped = false;               // bracketed reference and no survivor (wave-uniform): k0's value was never computed
        f32x4 rb0[NB], mu[NB];
        int rb0_mat = -1;                       // matrix index whose row-block 0 / means sit in rb0 / mu (wave-uniform)
        float tot_all[NG];
        int nev = 0;                            // clusters recorded for this tile (wave-uniform)
        auto record = [&](int k, float a) {     // table + running max / argmax bookkeeping for the owner lane
            if (tab_lds) ltab[k * WPTS] = a;
            else {
                if (valid) scr[(int64_t)k * sstride] = a;
                if (compact && screening) {
                    if (nev < A.lds_rows) { ltab[nev * WPTS] = a; if (lane == 0) slot_of[k] = (uint8_t)nev; }
                    ++nev;
                }
            }
            if (a != a) {
                if (!nan_seen || k < best) { nan_seen = true; best = k; }   // Julia's argmax: the first NaN wins
            } else if (a > m_run || (a == m_run && k < best && !nan_seen && m_run != -INFINITY)) {
                m_run = a;
                if (!nan_seen) best = k;
            }
        };
        // (ONE cluster, no table asked for: the draw returns index 0 whatever its value -- utils.jl:19-31 over one element -- so the value is not
        // computed: every fit from init_clusters = 1 spends its first `burnout` sweeps here, two evaluations per tile instead of three)
        const bool lone = !FAST && K == 1 && !a_tdf && !A.labels_only && !A.scratch_by_tile;
        if (!screening) {
            if (!lone) load_rb0<NB>(A.Rp, A.mup, rb0, mu, lane, g);
            for (int k = 0; k < (lone ? 0 : K); ++k) {
                const float *Rcur = A.Rp + (size_t)(3 * k) * MATSZ;
                const float *Rnext = (k + 1 < K) ? A.Rp + (size_t)(3 * (k + 1)) * MATSZ : nullptr;
                STAMP(q0);
                const float qs = quad_stream<NB, NG>(Rcur, Rnext, A.mup + (size_t)(3 * (k + 1)) * DP, rb0, mu, x, lane, g, true, tot_all, A.prio);
                ++nw_full;
                STAMP(q1);
                const float a = a_tdf ? A.cst[3 * k] - a_tdf[6 * k + 1] * log1pf(qs / a_tdf[6 * k])
                                      : __builtin_fmaf(-0.5f, qs, A.cst[3 * k]);
                record(k, a);
                STAMP(q2);
#ifdef DPMM_STAMPS
                T_quad += q1 - q0; T_epi += q2 - q1;
#endif
            }
        } else {
            // ---- screened label phase -----------------------------------------------------------------
            // (1) reference clusters: the previous labels of the wave's first and last point, evaluated in full;
            // (2) every other cluster: only the LAST 16-row block of y = R z (R upper triangular: it needs the last
            //     16 features only, 16 MFMAs); the sum over a lane's rows is a lower bound of q, so
            //     cst_k - q_lane/2 is an upper bound of a_k.  If it is below (reference - margin) for every point
            //     of the wave the cluster's probability is < e^-margin relative and it is skipped (a_k = -inf);
            // (3) survivors are evaluated in full.
            STAMP(q0);
            // rows of skipped clusters: the LDS-table draw visits evaluated clusters only (eval_bits), the global-table
            // draw scans all K rows and needs -inf there
            // (the global-table draw of a screened tile visits evaluated clusters only as well: no -inf rows to write)
            bool pvalid[NG];
#pragma unroll
            for (int n = 0; n < NG; ++n) pvalid[n] = wbase + 16 * n + ci < n_end;
            STAMP(qa);
            float bestn[NG];
#pragma unroll
            for (int n = 0; n < NG; ++n) bestn[n] = -INFINITY;
            auto full_eval = [&](int k, const float *Rnext, const float *mup_next) {
                const float qs = quad_stream<NB, NG>(A.Rp + (size_t)(3 * k) * MATSZ, Rnext, mup_next, rb0, mu, x, lane, g, true, tot_all, A.prio);
                ++nw_full;
                const float c = A.cst[3 * k];
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    const float an = __builtin_fmaf(-0.5f, tot_all[n], c);
                    if (an > bestn[n]) bestn[n] = an;
                }
                record(k, __builtin_fmaf(-0.5f, qs, c));
            };
            load_rb0<NB>(A.Rp + (size_t)(3 * k0) * MATSZ, A.mup + (size_t)(3 * k0) * DP, rb0, mu, lane, g);
            // (2a) scalar pre-screen: ||x - mu_k|| >= ||mu_k - mu_k0|| - ||x - mu_k0|| and q_k >= lam_k ||x - mu_k||^2.
            // Its operands are fetched / formed BEFORE the reference evaluation so that their latency hides behind it.
            STAMP(qb0);
            float rn[NG];
            const bool prescreen = !FAST && A.lam != nullptr;
            float pc_c = 0.f, pc_l = 0.f, pc_d = 0.f;      // lane j: constants of cluster j (first chunk of 64)
            if (prescreen) {
                if (lane < K) { pc_c = A.cst[3 * lane]; pc_l = A.lam[lane]; pc_d = A.mdist[(size_t)k0 * K + lane] * 0.99999f; }
#ifdef DPMM_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                STAMP(qb);
#ifdef DPMM_STAMPS
                T_i1 += qa - q0; T_i2 += qb - qa;
#endif
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    float part = 0.f;
#pragma unroll
                    for (int t = 0; t < NB; ++t) {
                        const f32x4 dz = x[n][t] - mu[t];              // mu = means of k0 (just loaded by load_rb0)
                        part = __builtin_fmaf(dz.x, dz.x, part); part = __builtin_fmaf(dz.y, dz.y, part);
                        part = __builtin_fmaf(dz.z, dz.z, part); part = __builtin_fmaf(dz.w, dz.w, part);
                    }
                    const f32x4 tot = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, part, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    rn[n] = sqrtf(tot[0]) * 1.00001f;
                }
            }
            STAMP(q0b);
            // the last row-block of the reference evaluation prefetches row-block 0 of k0's LEFT sub-cluster matrix:
            // on label-homogeneous waves that is the first matrix of the sub-label phase (rb0_mat tracks what rb0/mu hold)
            const float *Rl0 = ((!FAST && A.labels_only) || LSTORE) ? nullptr : A.Rp + (size_t)(3 * k0 + 1) * MATSZ;      // (LSTORE: no sub-label phase here, nothing to prefetch for it)
            const float *ml0 = A.mup + (size_t)(3 * k0 + 1) * DP;
            // Reference bracket (see ref_bracket): every point of the wave was in k0 -> bestn = the LOWER end of a certified bracket of
            // a_k0; nothing is recorded.  The Float32 evaluation follows behind the screens only if some cluster survives them.
            bool bracketed = false;
            if constexpr (NB == 4) {
                bool try_bracket = false;
                // (LIST: a span is here because niw_lean_kernel could not settle it against this very bracket -- the screens of a retry would run
                //  against the same thresholds: the exact reference value first, one evaluation.  Labels are the same with or without a bracket.
                //  What round 5 took for "a far outlier tile" in 45 % of the sweeps -- launches of 45 and 100 us -- were tiles with a uniform of
                //  exactly 0: gone since the uniforms are in (0, 1), dpmm_device.h u01; scripts/parts_trace.py: 24 of 24 sweeps hand on nothing.)
                if (!LIST && A.bracket && k1 == k0 && A.use_prev && (FAST || (A.tail != nullptr && !A.labels_only))) {
                    if (br_skip > 0) --br_skip;
                    else {
                        const int prevl0 = binv >= 0 ? (binv >> 1) : -1;
                        try_bracket = __ballot(valid && prevl0 != k0) == 0ull;      // every point of the wave was in k0
                    }
                }
                if (try_bracket) {
                    float qhi[NG];
                    ref_bracket<NG>(refb_records(A.tail, K) + (size_t)k0 * REFB_WORDS, x, mu, lane, qhi);
                    const float c0 = A.cst[3 * k0];
#pragma unroll
                    for (int n = 0; n < NG; ++n) bestn[n] = __builtin_fmaf(-0.5f, qhi[n], c0);
                    bracketed = true;
                    ++nw_br;                                                  // (48 bf16 matrix instructions, counted on their own: not Float32 matrix work)
                }
            }
            if (!bracketed) {
            full_eval(k0, k1 != k0 ? A.Rp + (size_t)(3 * k1) * MATSZ : Rl0, k1 != k0 ? A.mup + (size_t)(3 * k1) * DP : ml0);
            if (k1 != k0) full_eval(k1, Rl0, ml0);
            }
            // Further reference clusters: previous labels of the wave's points other than k0 / k1 -- a tile that covers a whole tiny
            // cluster (< 62 points) between two others.  Without them those points' reference is hopeless, no cluster can be excluded
            // for them and all K are evaluated in full: one such tile cost 30 evaluations (8 tile times) and, statically scheduled, ran
            // 8 % past the end of every other wave at N = 1e7.  Up to two more (four distinct labels per 64 points).
            int xr0 = -1, xr1 = -1;
            if (A.use_prev && !bracketed) {
                int prevl = binv >= 0 ? (binv >> 1) : -1;
                if ((unsigned)prevl >= (unsigned)K) prevl = -1;
                unsigned long long oth = __ballot(prevl >= 0 && prevl != k0 && prevl != k1);
                for (int e = 0; e < 2 && oth; ++e) {
                    const int kx = __builtin_amdgcn_readfirstlane(__shfl(prevl, __ffsll((long long)oth) - 1));
                    oth &= ~__ballot(prevl == kx);
                    load_rb0<NB>(A.Rp + (size_t)(3 * kx) * MATSZ, A.mup + (size_t)(3 * kx) * DP, rb0, mu, lane, g);
                    full_eval(kx, Rl0, ml0);                          // (prefetches the left sub-cluster's first fragments again: same state as after k0 / k1)
                    if (e == 0) xr0 = kx; else xr1 = kx;
                }
            }
            rb0_mat = ((!FAST && A.labels_only) || !Rl0) ? -1 : 3 * k0 + 1;
            if (pf_p >= 0 && A.use_prev) pf_bin = A.bins[pf_p];     // next tile's previous labels
            STAMP(r1);
            // (2a') VALU tail screen, lane = point: rows D-4..D-1 of y = R z need the last four features only
            // (R upper triangular), so q >= |T_k (x_tail - mu_tail)|^2 with the 4x4 tail factor T_k -- ~25 VALU
            // instructions per cluster for all 64 points, against 16 NG MFMAs for the 16-row screen below.
            const bool tailscr = FAST || A.tail != nullptr;
            f32x4 xt = (f32x4){0.f, 0.f, 0.f, 0.f};
            float my_best = -INFINITY;
            if (tailscr) {
                const int src = ci + 16 * A.tail_g;
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    f32x4 v;
                    v.x = __shfl(x[n][NB - 1].x, src); v.y = __shfl(x[n][NB - 1].y, src);
                    v.z = __shfl(x[n][NB - 1].z, src); v.w = __shfl(x[n][NB - 1].w, src);
                    if (g == n) { xt = v; my_best = bestn[n]; }
                }
            }
            const float my_thr = valid ? my_best - A.screen_margin : INFINITY;
            BallWave ball; ball.ok = false;
            if (tailscr && A.ball) ball = ball_of_wave(A.tail, K, k0, xt, my_thr, valid);
            // far mask: lane j owns cluster (chunk base + j).  The wave is reduced to its worst case first
            // (r_max = farthest point from mu_k0, best_min = lowest reference value), so one vector step tests 64 clusters:
            //   a_k(x) <= cst_k - lam_k/2 (||mu_k - mu_k0|| - r_max)_+^2  <  best_min - margin   for every point x of the wave.
            float r_max = 0.f, b_min = INFINITY;
            if (prescreen) {
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    if (pvalid[n]) { r_max = fmaxf(r_max, rn[n]); b_min = fminf(b_min, bestn[n]); }
                }
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {      // the 16 columns; the 4 row groups hold copies
                    r_max = fmaxf(r_max, __shfl_xor(r_max, off));
                    b_min = fminf(b_min, __shfl_xor(b_min, off));
                }
            }
            float ch_c = pc_c;                                // cst of cluster (farbase + lane) for the current chunk
            auto far_chunk = [&](int base) -> unsigned long long {
                const int j = base + lane;
                float cj = pc_c, lj = pc_l, dj = pc_d;
                if (base > 0) {
                    cj = lj = dj = 0.f;
                    if (j < K) { cj = A.cst[3 * j]; lj = A.lam[j]; dj = A.mdist[(size_t)k0 * K + j] * 0.99999f; }
                }
                ch_c = cj;
                const float dd = fmaxf(dj - r_max, 0.f);
                const bool far = (j < K) && lj > 0.f && (cj - 0.5f * lj * dd * dd < b_min - A.screen_margin);
                return __ballot(far);
            };
            // (2b) MFMA screen of the clusters that remain
            uint32_t *sv = surv_bits[tid >> 6];
            uint32_t *ev = eval_bits[tid >> 6];      // every cluster whose a_k is finite in the table (refs + survivors)
            if (lane < 32) { sv[lane] = 0u; ev[lane] = 0u; }
            if (lane == 0) {
                ev[k0 >> 5] |= 1u << (k0 & 31); ev[k1 >> 5] |= 1u << (k1 & 31);
                if (xr0 >= 0) ev[xr0 >> 5] |= 1u << (xr0 & 31);
                if (xr1 >= 0) ev[xr1 >> 5] |= 1u << (xr1 & 31);
            }
            const float margin = A.screen_margin;
            constexpr int LASTP = NP - 1, LB = NB - 1;
            STAMP(r1a);
            STAMP(r1b);
            // candidates = clusters that are neither references nor provably far; each gets the MFMA screen
            // (a two-stage pipeline of issue/finish was measured: no gain, and it spills)
            auto issue = [&](int k, f32x4 (&acc)[NG]) {
                f32x4 a, m4;
                if (A.screen_lds) {
                    a = *reinterpret_cast<const f32x4 *>(scrA + (size_t)k * 256 + lane * 4);
                    m4 = *reinterpret_cast<const f32x4 *>(scrM + (size_t)k * 16 + 4 * g);
                } else {
                    a = *reinterpret_cast<const f32x4 *>(A.Rp + ((size_t)(3 * k) * NP + LASTP) * 256 + lane * 4);
                    m4 = *reinterpret_cast<const f32x4 *>(A.mup + (size_t)(3 * k) * DP + 16 * LB + 4 * g);
                }
                f32x4 zz[NG];
#pragma unroll
                for (int n = 0; n < NG; ++n) { acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; zz[n] = x[n][LB] - m4; }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int n = 0; n < NG; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], zz[n][jj], acc[n], 0, 0, 0);
            };
            auto finish = [&](int k, float c, const f32x4 (&acc)[NG]) {
                ++nw_scr;
#ifdef DPMM_STAMPS
                ++N_scr;
#endif
                bool skip = true;
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    float ql = acc[n][0] * acc[n][0];
                    ql = __builtin_fmaf(acc[n][1], acc[n][1], ql);
                    ql = __builtin_fmaf(acc[n][2], acc[n][2], ql);
                    ql = __builtin_fmaf(acc[n][3], acc[n][3], ql);
                    const bool cond = !pvalid[n] || (__builtin_fmaf(-0.5f, ql, c) < bestn[n] - margin);
                    unsigned long long mk = __ballot(cond);     // a point is covered if ANY of its 4 row-group lanes proves the bound
                    mk |= mk >> 32;
                    mk |= mk >> 16;
                    skip = skip && ((mk & 0xFFFFull) == 0xFFFFull);
                }
                if (!skip && lane == 0) { sv[k >> 5] |= 1u << (k & 31); ev[k >> 5] |= 1u << (k & 31); }
            };
            for (int base = 0; base < K; base += 64) {
                unsigned long long cand = (K - base >= 64) ? ~0ull : ((1ull << (K - base)) - 1ull);
                if (prescreen) cand &= ~far_chunk(base);            // also loads ch_c for this chunk
                if (k0 >= base && k0 < base + 64) cand &= ~(1ull << (k0 - base));
                if (k1 >= base && k1 < base + 64) cand &= ~(1ull << (k1 - base));
                if (xr0 >= base && xr0 < base + 64) cand &= ~(1ull << (xr0 - base));
                if (xr1 >= base && xr1 < base + 64) cand &= ~(1ull << (xr1 - base));
                auto pop = [&]() -> int {
                    if (!cand) return -1;
                    const int b = __builtin_ctzll(cand);
                    cand &= cand - 1ull;
                    return base + b;
                };
                auto cst_of = [&](int k) -> float {
                    return prescreen ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ch_c), k - base)) : A.cst[3 * k];
                };
                f32x4 acc[NG];
                if (tailscr) {
                    if (ball.ok) cand &= ~ball_far(A.tail, K, base, lane, ball);
                    // pairs (2p, 2p+1) of the chunk that still hold a candidate; far clusters lose their candidate bit
#define DPMM_TAIL_PAIRS()                                                                                                            \
                    for (unsigned long long pend = cand; pend;) {                                                                \
                        const int sh = __builtin_ctzll(pend) & ~1;            /* (base is a multiple of 64: pair 2p sits at an even bit) */ \
                        const int pr = (base + sh) >> 1;                                                                         \
                        pend &= ~(3ull << sh);                                                                                   \
                        ++nw_tail;                                                                                               \
                        cand &= ~((unsigned long long)tail_pair_far(tail_load_pair(A.tail, pr), xt, my_thr) << sh);             \
                    }
                    // (DIR, (A.bf16scr & 2): the direction screen runs in FRONT of the tail pairs -- in the regime that switched it on they exclude
                    // nothing and cost 16 pair tests per tile; every 32nd sweep keeps the usual order and counts what the 4-row tests leave)
                    if constexpr (!DIR) { DPMM_TAIL_PAIRS() }
                    else if (!(A.bf16scr & 2)) { DPMM_TAIL_PAIRS() }
                    if constexpr (NB == 4) {
                        if (A.bf16scr) {
                            // bf16 screens of the remaining candidates, software-pipelined: the operands of the NEXT candidate (fragment, tail means,
                            // constant: dependent L2 / LDS / scalar loads) are requested before the current one is tested -- a screen is ~120 vector
                            // instructions and 8 short matrix instructions, the loads' latency was as long again
                            float thrb[NG];
                            f32x4 x3[NG];
#pragma unroll
                            for (int n = 0; n < NG; ++n) { thrb[n] = pvalid[n] ? bestn[n] - margin : INFINITY; x3[n] = x[n][LB]; }
                            // six or more candidates left by the 4-row tests: all of them at once, each along its own direction (direction_far); the
                            // number of candidates is what the library decides on for the NEXT sweep (without the screen it is the number of bottom screens)
                            if constexpr (DIR) {
                                const int nc = __builtin_popcountll(cand);
                                nw_cand += (unsigned)nc;                  // (direction first: before the tail pairs -- an upper bound of what they leave)
                                if (nc >= 6) {                            // (below: the candidates' own 16-row screens are cheaper, ~1/6 of this one each)
                                    ++nw_sp;
                                    cand &= ~direction_far<NG>(A.sp_frag + (size_t)k0 * SP_FRAG_WORDS, A.sp_cons + (size_t)k0 * SP_CONS_FLOATS,
                                                              A.mup + (size_t)(3 * k0) * DP, x, thrb, lane, g, K);
                                    nw_dcand += (unsigned)nc;
                                    nw_dexcl += (unsigned)(nc - __builtin_popcountll(cand));
                                }
                                if ((A.bf16scr & 2)) { DPMM_TAIL_PAIRS() }
                            }
                            const u32x4_t *Rb0 = reinterpret_cast<const u32x4_t *>(refb_records(A.tail, K));
                            auto loadk = [&](int k, u32x4_t &a, f32x4 &m4, float &ck) {
                                a = Rb0[(size_t)k * (REFB_WORDS / 4) + 64 * 5 + lane];
                                m4 = A.screen_lds ? *reinterpret_cast<const f32x4 *>(scrM + (size_t)k * 16 + 4 * g)
                                                  : *reinterpret_cast<const f32x4 *>(A.mup + (size_t)(3 * k) * DP + 16 * LB + 4 * g);
                                ck = cst_of(k);
                            };
                            u32x4_t a_c, a_n = (u32x4_t){0u, 0u, 0u, 0u};
                            f32x4 m_c, m_n = (f32x4){0.f, 0.f, 0.f, 0.f};
                            float c_c, c_n = 0.f;
                            int k = pop();
                            if (k >= 0) loadk(k, a_c, m_c, c_c);
                            while (k >= 0) {
                                const int kn = pop();
                                if (kn >= 0) loadk(kn, a_n, m_n, c_n);
                                ++nw_bb;
                                bool gone = bf16_bottom_excludes<NG>(a_c, x3, m_c, c_c, thrb);
                                if (!gone) {
                                    // passed the bottom rows: the top rows (block row 0 sees every feature) before any Float32 work
                                    f32x4 muk[4];
#pragma unroll
                                    for (int t = 0; t < 4; ++t) muk[t] = *reinterpret_cast<const f32x4 *>(A.mup + (size_t)(3 * k) * DP + 16 * t + 4 * g);
                                    ++nw_bt;
                                    gone = bf16_top_excludes<NG>(refb_records(A.tail, K) + (size_t)k * REFB_WORDS, x, muk, lane, c_c, thrb);
                                }
                                if (!gone) {
                                    issue(k, acc);
                                    finish(k, c_c, acc);
                                }
                                k = kn; a_c = a_n; m_c = m_n; c_c = c_n;
                            }
                        }
                    }
                    for (int k = pop(); k >= 0; k = pop()) {
                        issue(k, acc);
                        finish(k, cst_of(k), acc);
                    }
                } else {
                    for (int k = pop(); k >= 0; k = pop()) {
                        issue(k, acc);
                        finish(k, cst_of(k), acc);
                    }
                }
            }
            STAMP(r2);
            const int nwords = (K + 31) >> 5;
            if (bracketed) {
                uint32_t any = 0u;
                for (int w2 = 0; w2 < nwords; ++w2) any |= __builtin_amdgcn_readfirstlane(sv[w2]);
                if (any == 0u) {
                    // nobody can compete with k0 for any point of the wave: the draw returns k0 (or index 0 for u = 0) whatever the exact
                    // value is -- the one-cluster draw below never reads the table.  A finite stand-in keeps the bookkeeping of the draw.
                    m_run = 0.f; best = k0;
                    ref_skipped = true;
                    br_fail = 0;
                    if (Rl0) { load_rb0<NB>(Rl0, ml0, rb0, mu, lane, g); rb0_mat = 3 * k0 + 1; } else rb0_mat = -1;
                } else {
                    if (++br_fail >= 2) { br_fail = 0; br_skip = 8; }
                    load_rb0<NB>(A.Rp + (size_t)(3 * k0) * MATSZ, A.mup + (size_t)(3 * k0) * DP, rb0, mu, lane, g);      // (re-read: not kept across the bracket and the screens)
                    full_eval(k0, Rl0, ml0);             // the exact value: table entry for the draw, thresholds of the survivors' evaluations
                }
            }
            // (3) survivors, with one-matrix lookahead for the fragment prefetch
            int w = 0;
            uint32_t bits = __builtin_amdgcn_readfirstlane(sv[0]);
            auto next_surv = [&]() -> int {
                while (bits == 0u) {
                    ++w;
                    if (w >= nwords) return -1;
                    bits = __builtin_amdgcn_readfirstlane(sv[w]);
                }
                const int bpos = __builtin_ctz(bits);
                bits &= bits - 1u;
                return (w << 5) + bpos;
            };
            int kc = next_surv();
            if (kc >= 0) load_rb0<NB>(A.Rp + (size_t)(3 * kc) * MATSZ, A.mup + (size_t)(3 * kc) * DP, rb0, mu, lane, g);
            while (kc >= 0) {
                const int kn = next_surv();
                const float *Rn = kn >= 0 ? A.Rp + (size_t)(3 * kn) * MATSZ : Rl0;
                const float *mn = kn >= 0 ? A.mup + (size_t)(3 * kn) * DP : ml0;
                // evaluation with an exit after the first row block (see quad_stream<.., EARLY>): thresholds against the best value so far
                float thr4[NG];
#pragma unroll
                for (int n = 0; n < NG; ++n) thr4[n] = pvalid[n] ? bestn[n] - margin : INFINITY;
                const float c = A.cst[3 * kc];
                bool excl = false;
                const float qs = quad_stream<NB, NG, true>(A.Rp + (size_t)(3 * kc) * MATSZ, Rn, mn, rb0, mu, x, lane, g, true, tot_all, A.prio, c, thr4, &excl);
                if (excl) {
                    nw_scr += 4;                                              // (64 + 4 matrix instructions: four 16-row screens' worth)
                    if (lane == 0) ev[kc >> 5] &= ~(1u << (kc & 31));        // never recorded: the draw does not visit it
                    if (Rn) load_rb0<NB>(Rn, mn, rb0, mu, lane, g);
                } else {
                    ++nw_full;
#pragma unroll
                    for (int n = 0; n < NG; ++n) {
                        const float an = __builtin_fmaf(-0.5f, tot_all[n], c);
                        if (an > bestn[n]) bestn[n] = an;
                    }
                    record(kc, __builtin_fmaf(-0.5f, qs, c));
                }
                kc = kn;
            }
            STAMP(q1);
#ifdef DPMM_STAMPS
            T_init += q0b - q0; T_quad += r1 - q0b; T_epi += r2 - r1b; T_surv += q1 - r2; T_prep += r1a - r1; T_far += r1b - r1a;  // refs / K-loop / survivors / screen setup / (unused)
#endif
        }
        STAMP(s2);

        int z = 0;
        float u_sub = 0.f;
        bool nev1 = false;                     // exactly one cluster was evaluated for this tile, and it is k0 (wave-uniform)
        if (tab_lds && screening) {
            const uint32_t *evw = eval_bits[tid >> 6];
            int cntb = 0;
            for (int w = 0; w < ((K + 31) >> 5); ++w) cntb += __builtin_popcount(__builtin_amdgcn_readfirstlane(evw[w]));
            nev1 = cntb == 1;
        }
        // (the key is made opaque per tile: otherwise the compiler hoists the ten round keys out of the tile loop as 20 loop-invariant scalars,
        // spills them to vector lanes and fetches them back with 20 vector instructions per tile -- in the loop they are 20 scalar additions)
        uint32_t seed_lo = (uint32_t)A.seed, seed_hi = (uint32_t)(A.seed >> 32);
        asm volatile("" : "+s"(seed_lo), "+s"(seed_hi));
        if (valid) {
            const Philox4 r = philox4x32_10(((uint64_t)seed_hi << 32) | seed_lo, (uint64_t)(A.first_index + myp), A.epoch, STREAM_SWEEP);
            u_sub = u01(r.v[1]);
            if (A.final_argmax) {
                z = best;
            } else if (m_run == -INFINITY) {
                z = 0;
            } else if (ref_skipped) {
                z = (u01(r.v[0]) <= 0.f) ? 0 : k0;       // bracketed reference, nobody survived: one cluster of weight exp(0) = 1 (its value was never computed)
            } else if (tab_lds && screening && nev1) {
                // ONE evaluated cluster (the usual tile of well-separated data: every other cluster was excluded for the whole wave) and
                // its value is finite here: the row sum is exp(0) = 1, the scan stops at that cluster -- or at index 0 when u == 0, as the
                // full scan does (its threshold 0 is met before anything is added)
                z = (u01(r.v[0]) <= 0.f) ? 0 : k0;
            } else if (tab_lds && screening) {
                // skipped clusters hold -inf and contribute exact zeros: visit only the evaluated ones, in index order
                // (bit-identical to the full scan)
                const uint32_t *ev = eval_bits[tid >> 6];
                const int nw = (K + 31) >> 5;
                float s = 0.f;
                for (int w = 0; w < nw; ++w)
                    for (uint32_t bb = ev[w]; bb; bb &= bb - 1u) {
                        const int k = (w << 5) + __builtin_ctz(bb);
                        s += exp_det(nan_to_ninf(ltab[k * WPTS]) - m_run);
                    }
                const float t = u01(r.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                bool found = false;
                int last = 0;
                for (int w = 0; w < nw && !found; ++w)
                    for (uint32_t bb = ev[w]; bb; bb &= bb - 1u) {
                        const int k = (w << 5) + __builtin_ctz(bb);
                        last = k;
                        cw += exp_det(nan_to_ninf(ltab[k * WPTS]) - m_run);
                        if (!(cw < t)) { z = k; found = true; break; }
                    }
                // the full scan stops at the first k with cw >= t.  cw only changes at evaluated clusters, so if the
                // threshold was never reached the reference scan ends at K-1; if it was reached at an evaluated k it
                // is that k -- unless cw >= t already held BEFORE the first evaluated cluster (t == 0): then k = 0.
                if (t <= 0.f) z = 0;
                (void)last;
            } else if (screening) {
                // table in the global scratch (K beyond the LDS budget): evaluated clusters only, in index order, from the compact LDS copy
                // when all of them found a slot, else from the scratch rows -- same values, same order, same arithmetic as above
                const uint32_t *ev = eval_bits[tid >> 6];
                const int nw = (K + 31) >> 5;
                const bool from_lds = compact && nev <= A.lds_rows;
                auto val = [&](int k) -> float { return from_lds ? ltab[(int)slot_of[k] * WPTS] : scr[(int64_t)k * sstride]; };
                float s = 0.f;
                for (int w = 0; w < nw; ++w)
                    for (uint32_t bb = ev[w]; bb; bb &= bb - 1u) s += exp_det(nan_to_ninf(val((w << 5) + __builtin_ctz(bb))) - m_run);
                const float t = u01(r.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                bool found = false;
                for (int w = 0; w < nw && !found; ++w)
                    for (uint32_t bb = ev[w]; bb; bb &= bb - 1u) {
                        const int k = (w << 5) + __builtin_ctz(bb);
                        cw += exp_det(nan_to_ninf(val(k)) - m_run);
                        if (!(cw < t)) { z = k; found = true; break; }
                    }
                if (t <= 0.f) z = 0;
            } else if (tab_lds) {
                // same arithmetic, same order as the global-table path (and the CPU oracle); 4 table reads in flight
                float s = 0.f;
                int k = 0;
                for (; k + 4 <= K; k += 4) {
                    const float a0 = ltab[(k + 0) * WPTS], a1 = ltab[(k + 1) * WPTS], a2 = ltab[(k + 2) * WPTS], a3 = ltab[(k + 3) * WPTS];
                    s += exp_det(nan_to_ninf(a0) - m_run);
                    s += exp_det(nan_to_ninf(a1) - m_run);
                    s += exp_det(nan_to_ninf(a2) - m_run);
                    s += exp_det(nan_to_ninf(a3) - m_run);
                }
                for (; k < K; ++k) s += exp_det(nan_to_ninf(ltab[k * WPTS]) - m_run);
                const float t = u01(r.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                for (k = 0; k < K; ++k) {
                    cw += exp_det(nan_to_ninf(ltab[k * WPTS]) - m_run);
                    if (!(cw < t)) { z = k; break; }
                }
            } else {
                float s = 0.f;
                for (int k = 0; k < K; ++k) s += exp_det(nan_to_ninf(scr[(int64_t)k * sstride]) - m_run);
                const float t = u01(r.v[0]) * s;
                float cw = 0.f;
                z = K - 1;
                for (int k = 0; k < K; ++k) {
                    cw += exp_det(nan_to_ninf(scr[(int64_t)k * sstride]) - m_run);
                    if (!(cw < t)) { z = k; break; }
                }
            }
        }
        if (!FAST && A.labels_only) continue;
        STAMP(s3);
        if constexpr (LIST) {
            // A span handed on by niw_lean_kernel is finished HERE: the sub-labels of the labels just drawn, by the three-plane evaluation on
            // the x this wave still holds (b3_eval, niw_b3.h: the same device functions, operands and order as niw_sub_kernel -- the same
            // bits) -- one launch for the handed-on spans instead of two.  (This instantiation runs a few spans per sweep: its registers do not
            // matter.)
            if constexpr (NB == 4 && NG == 4) {
                float b0 = -INFINITY, b1 = -INFINITY;
                unsigned long long todo = __ballot(valid);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int kk = __builtin_amdgcn_readfirstlane(__shfl(z, leader));
                    todo &= ~__ballot(valid && z == kk);
                    f32x4 mk[4];
                    b3_mean(A.mup, kk, g, mk);
                    const B3Head H = b3_head(A.tail, A.cst, K, kk, lane, g);
                    B3Z Z;
                    b3_convert(x, mk, Z);
                    float bl, br;
                    b3_eval(A.tail, K, kk, Z, H, lane, g, bl, br);
                    if (z == kk) { b0 = bl; b1 = br; }
                    nw_b3l += 2;
                }
                if (valid) A.bins[myp] = 2 * z + draw2(b0, b1, u_sub);
            }
            nx_p = pf_p; nx_bin = pf_bin; nx_tile = tnext < tile_sent ? tnext : -1;
            l_pos = nl_pos; l_end = nl_end;
            continue;
        }
        if constexpr (LSTORE) {
            if (valid) A.bins[myp] = 2 * z;       // the label; niw_sub_kernel draws the sub-label (its uniform is the point's own: recomputed there)
            nx_p = pf_p; nx_bin = pf_bin; nx_tile = tnext < tile_sent ? tnext : -1;
            if (LIST) { l_pos = nl_pos; l_end = nl_end; }
            continue;
        }

        // sub-labels: walk the distinct labels of this wave (wave-uniform loop)
        float b0 = -INFINITY, b1 = -INFINITY;
        unsigned long long todo = __ballot(valid);
        auto next_label = [&]() -> int {
            if (!todo) return -1;
            const int leader = __ffsll((long long)todo) - 1;
            const int kk = __shfl(z, leader);
            todo &= ~__ballot(valid && z == kk);
            return kk;
        };
        int kcur = next_label();
        if (kcur >= 0 && rb0_mat != 3 * kcur + 1)
            load_rb0<NB>(A.Rp + (size_t)(3 * kcur + 1) * MATSZ, A.mup + (size_t)(3 * kcur + 1) * DP, rb0, mu, lane, g);
        while (kcur >= 0) {
            const int knext = next_label();
            const int jl = 3 * kcur + 1, jr = jl + 1, jn = 3 * (knext >= 0 ? knext : 0) + 1;
            const float cl = A.cst[jl], cr = A.cst[jr];      // (requested in front of the evaluations: fetched behind them, each was an exposed L2 round trip)
            const float bl = __builtin_fmaf(-0.5f, quad_stream<NB, NG>(A.Rp + (size_t)jl * MATSZ, A.Rp + (size_t)jr * MATSZ,
                                                                          A.mup + (size_t)jr * DP, rb0, mu, x, lane, g, true, tot_all, A.prio), cl);
            const float br = __builtin_fmaf(-0.5f, quad_stream<NB, NG>(A.Rp + (size_t)jr * MATSZ,
                                                                          knext >= 0 ? A.Rp + (size_t)jn * MATSZ : nullptr,
                                                                          A.mup + (size_t)jn * DP, rb0, mu, x, lane, g, true, tot_all, A.prio), cr);
            if (valid && z == kcur) { b0 = bl; b1 = br; }
            nw_full += 2;
            kcur = knext;
        }
        if (valid) A.bins[myp] = 2 * z + draw2(b0, b1, u_sub);
        nx_p = pf_p; nx_bin = pf_bin; nx_tile = tnext < nwtiles ? tnext : -1;
        STAMP(s4);
#ifdef DPMM_STAMPS
        T_draw += s3 - s2; T_p2 += s4 - s3; T_tot += s4 - s0; T_x += s1 - s0; ++ntile;
        if (s4 - s0 > T_long) { T_long = s4 - s0; T_longat = s0; }       // longest tile of this wave and when it began
        if (!T_firstd) T_firstd = s0;
        T_lastd = s4;
#endif
    }
    if (A.work && lane == 0) {      // one slot per wave, no atomics (see the LDS-staged kernel)
        unsigned long long *slot = A.work + DPMM_WORK_SLOTS + (size_t)wave_id * DPMM_WORK_PER_WAVE;      // (accumulates over launches; cleared by the reader)
        slot[0] += nw_tiles; slot[1] += nw_full; slot[2] += nw_scr; slot[3] += nw_tail; slot[4] += nw_br; slot[5] += nw_bb; slot[6] += nw_bt; slot[7] += nw_sp;
        if constexpr (LIST) slot[7] += (unsigned long long)nw_b3l << 32;
    }
    if (A.need && lane == 0) {                // (this launch only: plain store into the host's pinned block)
        const unsigned nc = DIR ? nw_cand : nw_bb;
        uint32_t word = ((nc < 65535u ? nc : 65535u) << 16) | (nw_tiles < 32767u ? nw_tiles : 32767u);
        if constexpr (DIR) { if ((A.bf16scr & 2)) word = (word & 0xFFFF7FFFu) | 0x8000u; }      // (bit 15: counted in front of the tail pairs, an upper bound)
        A.need[2 * wave_id] = word;
        if constexpr (DIR) A.need[2 * wave_id + 1] = ((nw_dexcl < 65535u ? nw_dexcl : 65535u) << 16) | (nw_dcand < 65535u ? nw_dcand : 65535u);
    }
#ifdef DPMM_STAMPS
    if (lane == 0 && A.dbg) {
        unsigned long long *d = A.dbg + wave_id * 16;
        d[0] = T_x; d[1] = T_quad; d[2] = T_prep; d[3] = T_epi; d[4] = T_surv; d[5] = T_draw; d[6] = T_p2; d[7] = T_tot;
        d[8] = ntile; d[9] = N_tail; d[10] = N_scr; d[11] = T_long; d[12] = T_init; d[13] = T_longat; d[14] = T_firstd; d[15] = T_lastd;
    }
#endif
}

#undef DPMM_TAIL_PAIRS

template <int NB, int NG, int OCC>
static hipError_t launch_direct(const NiwSweepArgs &a, int grid, hipStream_t s) {
    // a_k table in LDS: the CU's 160 KiB split over OCC resident workgroups of 4 waves
    NiwSweepArgs b = a;
    const int budget_rows = (int)((160 * 1024 / OCC - 512) / (4 * 16 * NG * sizeof(float)));
    // K beyond the budget: min(64, budget / 2) compact slots per wave for the evaluated clusters of a screened tile (+ K slot bytes per wave)
    b.lds_rows = a.K <= budget_rows ? a.K : (a.scratch_by_tile ? 0 : std::min(64, budget_rows / 2));
    size_t lds_bytes = (size_t)b.lds_rows * 4 * 16 * NG * sizeof(float);
    const size_t screen_bytes = (size_t)a.K * (256 + 16) * sizeof(float);
    const size_t slot_bytes = a.K > budget_rows ? 4 * (size_t)((a.K + 3) & ~3) : 0;
    b.screen_lds = (NB >= 2 && a.screen_margin > 0.f && lds_bytes + screen_bytes + slot_bytes + 1024 <= (size_t)(160 * 1024 / OCC)) ? 1 : 0;
    if (b.screen_lds) lds_bytes += screen_bytes;
    lds_bytes += slot_bytes;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
        attr_set = true;
    }
    // bit 2 of bf16scr (NB = 4, set by run_sweep while the bf16 sub-label evaluation is active): the LSTORE instantiations -- labels only, stored;
    // b.tdf then carries the tile list (or null: all tiles), never Student-t parameters
    const bool lstore = NB == 4 && (a.bf16scr & 4) != 0;
    b.bf16scr &= 3;
    const bool fast = NB >= 2 && b.screen_margin > 0.f && (lstore || !b.tdf) && !b.scratch_by_tile && !b.labels_only && b.K > 1 && b.lam == nullptr &&
                      b.tail != nullptr && b.lds_rows >= b.K && !b.final_argmax;
    if constexpr (NB == 4) {
        if (lstore) {
            static bool attr_ls = false;
            if (!attr_ls) {
                hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
                hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
                hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
                hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
                hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
                hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
                attr_ls = true;
            }
            if (b.tdf != nullptr) {             // a list of spans (behind niw_lean_kernel): with the direction screen while its tables exist (round 6: the lean kernel
                                                //  runs in that regime too, and the spans it hands on there keep dozens of candidates behind the 4-row tests)
                if (fast && b.sp_frag && b.sp_cons && b.bf16scr && b.K <= SP_MAXK) DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC, true, true, true, true>), dim3(grid), dim3(256), lds_bytes, s, b);
                else if (fast) DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC, true, false, true, true>), dim3(grid), dim3(256), lds_bytes, s, b);
                else DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC, false, false, true, true>), dim3(grid), dim3(256), lds_bytes, s, b);
                return hipGetLastError();
            }
            if (fast && b.sp_frag && b.sp_cons && b.bf16scr && b.K <= SP_MAXK) DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC, true, true, true>), dim3(grid), dim3(256), lds_bytes, s, b);
            else if (fast) DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC, true, false, true>), dim3(grid), dim3(256), lds_bytes, s, b);
            else DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC, false, false, true>), dim3(grid), dim3(256), lds_bytes, s, b);
            return hipGetLastError();
        }
        if (fast && b.sp_frag && b.sp_cons && b.bf16scr && b.K <= SP_MAXK) {
            static bool attr_dir = false;
            if (!attr_dir) {
                hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
                attr_dir = true;
            }
            DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC, true, true>), dim3(grid), dim3(256), lds_bytes, s, b);
            return hipGetLastError();
        }
    }
    if (fast) {
        static bool attr_fast = false;
        if (!attr_fast) {
            hipFuncSetAttribute((const void *)niw_sweep_direct_kernel<NB, NG, OCC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC);
            attr_fast = true;
        }
        DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC, true>), dim3(grid), dim3(256), lds_bytes, s, b);
    } else {
        DPMM_LAUNCH((niw_sweep_direct_kernel<NB, NG, OCC>), dim3(grid), dim3(256), lds_bytes, s, b);
    }
    return hipGetLastError();
}

template <int NB, int NG, int CH>
static hipError_t launch_cfg(const NiwSweepArgs &a, int grid, hipStream_t s) {
    DPMM_LAUNCH((niw_sweep_kernel<NB, NG, CH>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

int niw_occupancy(int NB) { return NB <= 8 ? 2 : 1; }

int niw_tile_points(int NB) {
    switch (NB) {
        case 4: return 256;
        case 1: case 2: return 256;
        case 8: return 128;
        default: return 128;
    }
}

hipError_t launch_niw_sweep(int NB, const NiwSweepArgs &a, int grid, hipStream_t s) {
    switch (NB) {
        case 1: return launch_direct<1, 4, 4>(a, grid, s);
        case 2: return launch_direct<2, 4, 3>(a, grid, s);
        case 4: return launch_direct<4, 4, 2>(a, grid, s);
        case 8: return launch_cfg<8, 2, 2>(a, grid, s);
        case 16: return launch_cfg<16, 2, 1>(a, grid, s);
        default: return hipErrorInvalidValue;
    }
}

// Diagnostic (dpmm_debug_ref_bracket): for every point of the shard, the bracket's upper end q_hi and the Float32 quadratic form q of
// cluster k's cluster-level factor -- the SAME device functions (ref_bracket, quad_stream), operand images and register layout as the
// sweep kernel, storage order, one wave per 64 points.  `c_override` > 0 replaces REFB_C (a test shows that round 3's constant fails).
__global__ __launch_bounds__(256) void niw_refb_debug_kernel(const float *__restrict__ X, int64_t ldx, int64_t n, const float *__restrict__ Rm,
                                                             const float *__restrict__ mup, const uint32_t *__restrict__ Rb, float c_override,
                                                             float *__restrict__ qhi_out, float *__restrict__ q_out) {
    constexpr int NB = 4, NG = 4;
    const int lane = threadIdx.x & 63, ci = lane & 15, g = lane >> 4;
    const int64_t wbase = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (wbase >= n) return;
    f32x4 x[NG][NB];
#pragma unroll
    for (int nn = 0; nn < NG; ++nn) {
        const int64_t p = wbase + 16 * nn + ci;
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int e = 16 * t + 4 * g;
            x[nn][t] = (p < n && e < ldx) ? *reinterpret_cast<const f32x4 *>(X + p * ldx + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 rb0[NB], mu[NB];
    load_rb0<NB>(Rm, mup, rb0, mu, lane, g);
    float qhi[NG], tot_all[NG];
    ref_bracket<NG>(Rb, x, mu, lane, qhi, c_override > 0.f ? c_override : REFB_C);
    quad_stream<NB, NG>(Rm, nullptr, mup, rb0, mu, x, lane, g, true, tot_all, 0);
#pragma unroll
    for (int nn = 0; nn < NG; ++nn) {
        const int64_t p = wbase + 16 * nn + ci;
        if (g == nn && p < n) { qhi_out[p] = qhi[nn]; q_out[p] = tot_all[nn]; }
    }
}
hipError_t launch_niw_refb_debug(const NiwSweepArgs &a, int k, float c_override, float *qhi_out, float *q_out, hipStream_t s) {
    constexpr int NP = 10, MATSZ = NP * 256, DP = 64;
    const unsigned grid = (unsigned)((a.n + 255) / 256);
    if (grid == 0) return hipSuccess;
    DPMM_LAUNCH(niw_refb_debug_kernel, dim3(grid), dim3(256), 0, s, a.X, a.ldx, a.n, a.Rp + (size_t)(3 * k) * MATSZ, a.mup + (size_t)(3 * k) * DP,
                refb_records(a.tail, a.K) + (size_t)k * REFB_WORDS, c_override, qhi_out, q_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Parameter packing: raw R (upper triangular, stored PACKED: row r holds columns r..D-1 at offset r D - r (r-1)/2 of a
// D (D+1)/2-float record per matrix -- half the bytes of the full square cross the host link) -> MFMA A-fragment image.
// Rp[j][pair(bi,t)][lane][jj] = R[16 bi + (lane & 15)][16 t + 4 (lane >> 4) + jj], zero beyond D
// and below the diagonal.  mup[j][DP] = mu zero-padded.
// `slot` (nullable): packed matrix j = 3k+w is read from source row 3*slot[k]+w (the host keeps a cluster's rows in place for
// life and only re-orders this map when clusters are removed); cst is always in cluster order.
__device__ __forceinline__ size_t src_row(const int32_t *__restrict__ slot, int j) {
    return slot ? (size_t)(3 * slot[j / 3] + j % 3) : (size_t)j;
}
__device__ __forceinline__ size_t tri_off(int D, int row, int col) {      // col >= row
    return (size_t)row * D - (size_t)row * (row - 1) / 2 + (size_t)(col - row);
}
__global__ void niw_pack_kernel(const float *__restrict__ R, const float *__restrict__ mu, float *__restrict__ Rp,
                                float *__restrict__ mup, int D, int NB, int nmat, float *__restrict__ tail, const float *__restrict__ cst,
                                const int32_t *__restrict__ slot, float *__restrict__ cst_out, unsigned long long *__restrict__ work) {
    const int NP = NB * (NB + 1) / 2;
    // riders (save two launches per sweep): the additive constants move from the parameter image to where the sweep kernels read
    // them, and the executed-work counters of the next sweep start at zero
    if (cst_out) for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nmat; e += (int64_t)gridDim.x * blockDim.x) cst_out[e] = cst[e];
    if (work && blockIdx.x == 0 && threadIdx.x < 8) { work[threadIdx.x] = 0ull; work[8 + 16 * threadIdx.x] = 0ull; }     // tile queue heads ([4]; [8 + 16 q])
    const int DP = 16 * NB;
    const size_t TRI = (size_t)D * (D + 1) / 2;
    const int64_t total = (int64_t)nmat * NP * 256;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int jj = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        const int64_t pj = e >> 8;
        const int pair = (int)(pj % NP);
        const int j = (int)(pj / NP);
        int bi = 0, rem = pair;
        while (rem >= NB - bi) { rem -= NB - bi; ++bi; }
        const int t = bi + rem;
        const int row = 16 * bi + (lane & 15);
        const int col = 16 * t + 4 * (lane >> 4) + jj;
        float v = 0.f;
        if (row < D && col < D && col >= row) v = R[src_row(slot, j) * TRI + tri_off(D, row, col)];
        Rp[e] = v;
    }
    const int64_t totmu = (int64_t)nmat * DP;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < totmu; e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % DP);
        const int j = (int)(e / DP);
        mup[e] = d < D ? mu[src_row(slot, j) * D + d] : 0.f;
    }
    // Tail records of the cluster-level matrices 3k, stored for cluster PAIRS: tail[pair][q][c], c = 0 / 1 for cluster 2 pair + c,
    // q = { T00 T01 T02 T03 | T11 T12 T13 T22 | T23 T33 m0 m1 | m2 m3 cst_k 0 },  T = R[D-4:D, D-4:D], m = mu[D-4:D].
    // A padding cluster (odd K) gets cst = -inf: it is "far" for everybody and its candidate bit does not exist anyway.
    if (tail && D >= 4) {
        const int f0 = D - 4, K = nmat / 3, NPR = (K + 1) / 2;
        for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)NPR * 32; e += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)(e & 1), q = (int)((e >> 1) & 15), k = 2 * (int)(e >> 5) + c;
            const size_t j = k < K ? src_row(slot, 3 * k) : 0;
            const int tr[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, tc[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
            float v = 0.f;
            if (k >= K) v = (q == 14) ? -INFINITY : 0.f;
            else if (q < 10) v = R[j * TRI + tri_off(D, f0 + tr[q], f0 + tc[q])];
            else if (q < 14) v = mu[j * D + f0 + (q - 10)];
            else if (q == 14) v = cst[3 * k];
            tail[e] = v;
        }
        // per-cluster records of the ball test (lane = cluster): { m | T row 0 | T11 T12 T13 T22 | T23 T33 |T|_F cst }
        float *ball = tail + 32 * NPR;
        for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)K * 16; e += (int64_t)gridDim.x * blockDim.x) {
            const int q = (int)(e & 15), k = (int)(e >> 4);
            const size_t j = src_row(slot, 3 * k);
            const int tr[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, tc[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
            float v;
            if (q < 4) v = mu[j * D + f0 + q];
            else if (q < 14) v = R[j * TRI + tri_off(D, f0 + tr[q - 4], f0 + tc[q - 4])];
            else if (q == 14) {
                float t10[10];
                for (int i = 0; i < 10; ++i) t10[i] = R[j * TRI + tri_off(D, f0 + tr[i], f0 + tc[i])];
                v = tail_opnorm_bound(t10);
            } else v = cst[3 * k];
            ball[e] = v;
        }
        // bf16 image of the cluster-level factors for the reference bracket (refb_map), D in 33 .. 64 only (NB = 4)
        if (NB == 4) {
            uint32_t *refb = reinterpret_cast<uint32_t *>(ball + 16 * (size_t)K);
            for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)K * REFB_WORDS; e += (int64_t)gridDim.x * blockDim.x) {
                const int k = (int)(e / REFB_WORDS);
                int row, c0, c1;
                refb_map((int)(e % REFB_WORDS), row, c0, c1);
                const size_t j = src_row(slot, 3 * k);
                const float v0 = (row < D && c0 < D && c0 >= row) ? R[j * TRI + tri_off(D, row, c0)] : 0.f;
                const float v1 = (row < D && c1 < D && c1 >= row) ? R[j * TRI + tri_off(D, row, c1)] : 0.f;
                refb[e] = bf16_rne_bits(v0) | (bf16_rne_bits(v1) << 16);
            }
        }
    }
}

// Screening constants of the cluster-level distributions (rows 3k), one workgroup per cluster:
//   lam[k]   = 0.9 / ||R_k^-1||_F^2  <=  sigma_min(R_k)^2 = lambda_min(Sigma_k^-1)   (R upper triangular, row-major [D][D])
//   dist[k][j] = || mu_k - mu_j ||_2
// Thread c solves R v = e_c by back substitution (column c of R^-1); v lives in LDS.
__global__ __launch_bounds__(256) void niw_screen_prep_kernel(const float *__restrict__ R, const float *__restrict__ mu,
                                                              int D, int K, float *__restrict__ lam, float *__restrict__ dist,
                                                              const int32_t *__restrict__ slot) {
    extern __shared__ float sh[];            // Rk [D*D] | v [D][D+1] | red[256]
    float *Rk = sh, *v = sh + (size_t)D * D, *red = v + (size_t)D * (D + 1);
    const int k = blockIdx.x, tid = threadIdx.x;
    const float *Rg = R + src_row(slot, 3 * k) * ((size_t)D * (D + 1) / 2);
    for (int e = tid; e < D * D; e += blockDim.x) { const int r = e / D, cc = e % D; Rk[e] = cc >= r ? Rg[tri_off(D, r, cc)] : 0.f; }
    __syncthreads();
    float ss = 0.f;
    for (int c = tid; c < D; c += blockDim.x) {
        float *vc = v + (size_t)c * (D + 1);
        for (int i = c; i >= 0; --i) {
            float acc = (i == c) ? 1.f : 0.f;
            for (int t = i + 1; t <= c; ++t) acc -= Rk[(size_t)i * D + t] * vc[t];
            const float val = acc / Rk[(size_t)i * D + i];
            vc[i] = val;
            ss += val * val;
        }
    }
    red[tid] = ss;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const float f = red[0];
        lam[k] = (f > 0.f && f == f && f < INFINITY) ? 0.9f / f : 0.f;   // 0 disables the bound for this cluster
    }
    const float *mk = mu + src_row(slot, 3 * k) * D;
    for (int j = tid; j < K; j += blockDim.x) {
        const float *mj = mu + src_row(slot, 3 * j) * D;
        float d2 = 0.f;
        for (int d = 0; d < D; ++d) { const float t = mk[d] - mj[d]; d2 += t * t; }
        dist[(size_t)k * K + j] = sqrtf(d2);
    }
}

hipError_t launch_niw_screen_prep(const float *R, const float *mu, int D, int K, float *lam, float *dist, const int32_t *slot, hipStream_t s) {
    const size_t lds = sizeof(float) * ((size_t)D * D + (size_t)D * (D + 1) + 256);
    DPMM_LAUNCH(niw_screen_prep_kernel, dim3(K), dim3(256), lds, s, R, mu, D, K, lam, dist, slot);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// bf16 image of the cluster-level factors for ref_bracket_big (NB = 8, 16), from the Float32 fragment image both pack kernels write:
// fragment (bi, sl), lane (i, g), dword d = k-slots 2 d, 2 d + 1 = features 32 sl + 4 g + 2 d (+1) for d < 2 -- block (bi, t = 2 sl) of the
// Float32 image, same lane, elements 2 d, 2 d + 1 -- and 32 sl + 16 + 4 g + 2 (d - 2) (+1) for d >= 2 -- block (bi, 2 sl + 1), elements
// 2 d - 4, 2 d - 3; a block left of the diagonal (t < bi) is not stored: zero.
__global__ void niw_refb_big_kernel(const float *__restrict__ Rp, int NB, int K, uint32_t *__restrict__ out) {
    const int NP = NB * (NB + 1) / 2, NSL = NB / 2;
    int nfr = 0;
    for (int bi = 0; bi < NB; ++bi) nfr += NSL - bi / 2;
    const int64_t total = (int64_t)K * nfr * 256;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e & 3), ln = (int)((e >> 2) & 63);
        const int64_t fk = e >> 8;
        const int f = (int)(fk % nfr), k = (int)(fk / nfr);
        int bi = 0, rem = f;
        while (rem >= NSL - bi / 2) { rem -= NSL - bi / 2; ++bi; }
        const int sl = bi / 2 + rem, t = 2 * sl + (d >> 1), j = 2 * (d & 1);
        float v0 = 0.f, v1 = 0.f;
        if (t >= bi) {
            const int pair = bi * NB - bi * (bi - 1) / 2 + (t - bi);            // pair_base(bi) + (t - bi)
            const float *src = Rp + ((size_t)(3 * k) * NP + pair) * 256 + ln * 4 + j;
            v0 = src[0]; v1 = src[1];
        }
        out[e] = bf16_rne_bits(v0) | (bf16_rne_bits(v1) << 16);
    }
}
hipError_t launch_niw_bracket_big(int NB, const NiwSweepArgs &a, const uint32_t *refb, uint32_t *tile_flag, float *aref, hipStream_t s) {
    const int ntiles = (int)((a.n + 127) / 128);       // 4 waves x 32 points, as niw_sweep_kernel<8 | 16, 2, .>
    if (ntiles == 0) return hipSuccess;
    if (NB == 16) DPMM_LAUNCH((niw_bracket_big_kernel<16, 2>), dim3(ntiles), dim3(256), 0, s, a.X, a.ldx, a.n, a.order, a.order_total, a.bins, a.K, a.mup, a.cst, refb, tile_flag, aref);
    else if (NB == 8) DPMM_LAUNCH((niw_bracket_big_kernel<8, 2>), dim3(ntiles), dim3(256), 0, s, a.X, a.ldx, a.n, a.order, a.order_total, a.bins, a.K, a.mup, a.cst, refb, tile_flag, aref);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_niw_refb_big(const float *Rp, int NB, int K, uint32_t *out, hipStream_t s) {
    if ((NB != 8 && NB != 16) || K < 1) return hipErrorInvalidValue;
    DPMM_LAUNCH(niw_refb_big_kernel, dim3(256), dim3(256), 0, s, Rp, NB, K, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Tables of the direction screen (direction_far), from the images the sweep itself reads (D in 33 .. 64: NB = 4, NP = 10, DP = 64).
// One workgroup per (cluster k, 16 reference clusters k0), sixteen threads per k0: R_k (the Float32 values of the fragment image) in LDS,
//     d = mu_k0 - mu_k,   y = R_k d  (Float64 sums),   b = |y|,   u = y / b,   w = R_k' u,
// w as the bf16 A-operand row of cluster k in k0's fragment table (k-slots in the order of the bracket's image, refb_map), and the constants
//     B = b,      E = (c + 1e-6) |w| (1 + 1e-5),      cst_k.
// A pair without a direction (k = k0, b not finite or zero) and the rows of clusters that do not exist get E = inf: t = 0, nothing excluded.
__global__ __launch_bounds__(256) void niw_direction_kernel(const float *__restrict__ Rp, const float *__restrict__ mup, const float *__restrict__ cst,
                                                            int D, int K, uint32_t *__restrict__ frag, float *__restrict__ cons) {
    // workgroup (k, y): cluster k against the 16 reference clusters k0 = 16 y .. 16 y + 15, sixteen threads per k0 (four rows / columns each)
    __shared__ float Rk[64 * 65];              // R_k, row stride 65
    __shared__ float dl[16][65];               // d of the workgroup's k0, then u, then w
    __shared__ float muk[64];
    const int k = blockIdx.x, q0 = 16 * blockIdx.y, tid = threadIdx.x;
    const float *img = Rp + (size_t)(3 * k) * 2560;
    for (int e = tid; e < 64 * 65; e += 256) Rk[e] = 0.f;
    if (tid < 64) muk[tid] = mup[(size_t)(3 * k) * 64 + tid];
    __syncthreads();
    for (int e = tid; e < 2560; e += 256) {
        const int jj = e & 3, ln = (e >> 2) & 63, pair = e >> 8;
        int bi = 0, rem = pair;
        while (rem >= 4 - bi) { rem -= 4 - bi; ++bi; }
        Rk[(16 * bi + (ln & 15)) * 65 + 16 * (bi + rem) + 4 * (ln >> 4) + jj] = img[e];
    }
    for (int e = tid; e < 16 * 64; e += 256) {
        const int ql = e >> 6, c = e & 63;
        dl[ql][c] = q0 + ql < K ? mup[(size_t)(3 * (q0 + ql)) * 64 + c] - muk[c] : 0.f;
    }
    __syncthreads();
    const int ql = tid >> 4, p = tid & 15, k0 = q0 + ql;       // rows / columns 4 p .. 4 p + 3 of reference cluster k0's pair
    double y[4];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * p + i;
        double acc = 0.0;
        for (int c = r; c < 64; ++c) acc = __builtin_fma((double)Rk[r * 65 + c], (double)dl[ql][c], acc);      // (upper triangular)
        y[i] = acc;
        ss = __builtin_fma(acc, acc, ss);
    }
    for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o);
    const double b = sqrt(ss);
    const bool live = k0 < K && k0 != k && b > 0.0 && b < 1e30;
    __syncthreads();                             // every d has been read
    const double ib = live ? 1.0 / b : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) dl[ql][4 * p + i] = (float)(y[i] * ib);       // u (Float32: its norm is 1 within 1e-6)
    __syncthreads();
    float w[4];
    double ww = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * p + i;
        double acc = 0.0;
        for (int r = 0; r <= c; ++r) acc = __builtin_fma((double)Rk[r * 65 + c], (double)dl[ql][r], acc);
        w[i] = (float)acc;
        ww = __builtin_fma(acc, acc, ww);
    }
    for (int o = 1; o < 16; o <<= 1) ww += __shfl_xor(ww, o);
    __syncthreads();                             // every u has been read
#pragma unroll
    for (int i = 0; i < 4; ++i) dl[ql][4 * p + i] = live ? w[i] : 0.f;
    __syncthreads();
    if (p == 0 && k0 < K) {                      // constants of cluster k in reference cluster k0's table
        float *co = cons + (size_t)k0 * SP_CONS_FLOATS;
        const float nw = (float)sqrt(ww) * 1.00001f;
        co[k] = live ? (float)b : 0.f;
        co[64 + k] = live ? (REFB_C + 1e-6f) * nw : INFINITY;          // (inf: t = 0, nothing excluded along a direction that does not exist)
        co[128 + k] = cst[3 * k];
    }
    if (k == 0 && tid < 64 * 3) {                // rows of clusters that do not exist: never excluded (they have no candidate bit either)
        for (int q = q0; q < min(K, q0 + 16); ++q) {
            float *co = cons + (size_t)q * SP_CONS_FLOATS;
            const int a = tid / 64, j = tid % 64;
            if (j >= K) co[64 * a + j] = a == 1 ? INFINITY : (a == 2 ? INFINITY : 0.f);
        }
    }
    // row (k & 15) of block (k >> 4) in the fragment table of each of the workgroup's k0: [blk][sl][lane = (i, g)][4 dwords]
    const int blk = k >> 4, irow = k & 15;
    const int nq = min(16, K - q0);
    for (int e = tid; e < nq * 2 * 4 * 4; e += 256) {         // (k0, sl, g, d)
        const int d = e & 3, g = (e >> 2) & 3, sl = (e >> 4) & 1, qq = e >> 5, j0 = 2 * d;
        const int col = 32 * sl + (j0 < 4 ? 4 * g + j0 : 16 + 4 * g + (j0 - 4));
        frag[(size_t)(q0 + qq) * SP_FRAG_WORDS + (((blk * 2 + sl) * 64) + (16 * g + irow)) * 4 + d] = bf16_rne_bits(dl[qq][col]) | (bf16_rne_bits(dl[qq][col + 1]) << 16);
    }
    if (k == K - 1) {
        for (int e = tid; e < nq * 2 * 4 * 4 * 16; e += 256) {  // zero rows K .. 16 ceil(K / 16) - 1 of the last block in each of these k0's tables
            const int d = e & 3, g = (e >> 2) & 3, sl = (e >> 4) & 1, rr = (e >> 5) & 15, qq = e >> 9;
            if (rr > irow) frag[(size_t)(q0 + qq) * SP_FRAG_WORDS + (((blk * 2 + sl) * 64) + (16 * g + rr)) * 4 + d] = 0u;
        }
    }
}
hipError_t launch_niw_direction(const float *Rp, const float *mup, const float *cst, int D, int K, uint32_t *frag, float *cons, hipStream_t s) {
    if (K < 1 || K > SP_MAXK || D < 33 || D > 64) return hipErrorInvalidValue;
    DPMM_LAUNCH(niw_direction_kernel, dim3(K, (K + 15) / 16), dim3(256), 0, s, Rp, mup, cst, D, K, frag, cons);
    return hipGetLastError();
}

hipError_t launch_niw_pack(const float *R, const float *mu, float *Rp, float *mup, int D, int NB, int nmat, float *tail, const float *cst,
                           const int32_t *slot, float *cst_out, unsigned long long *work, hipStream_t s) {
    DPMM_LAUNCH(niw_pack_kernel, dim3(512), dim3(256), 0, s, R, mu, Rp, mup, D, NB, nmat, tail, cst, slot, cst_out, work);
    return hipGetLastError();
}

}  // namespace dpmm
