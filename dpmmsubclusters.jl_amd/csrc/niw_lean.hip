// niw_lean.hip -- the bf16 three-plane evaluation of the SUB-CLUSTER quadratic forms (D in 33 .. 64) and the kernels built on it.
//
// Stands in for (reference paths relative to the reference checkout):
//   sample_sub_clusters_worker! / create_subclusters_labels!   src/local_clusters_actions.jl:70-95
//   log_likelihood!(::mv_gaussian)                              src/distributions/mv_gaussian.jl:21-25
//   and, for the tiles the cheap screens settle completely, sample_labels_worker! (src/local_clusters_actions.jl:112-134)
//
// Why a file of its own.  Every FP32-input matrix instruction holds the SIMD's vector issue port for its whole duration
// (scripts/microbench/mfma_shapes_issue.hip: a v_fma_f32 of the same wave behind a v_mfma_f32_16x16x4_f32 adds its full 5-9 cycles to the
// 32, the other wave of the SIMD gets one vector instruction per one to four of them); v_mfma_f32_16x16x32_bf16 hides two fillers
// completely and lets the other wave issue at full rate.  The D <= 64 sweep spent 10.5 k of a tile's 23 k SIMD cycles in the two Float32
// sub-cluster evaluations.  On the bf16 pipe they are 4.6 k cycles that overlap vector work -- but their operands (96 registers for
// the planes of z, 48 of fragments: 230 registers for the phase alone) do not fit beside what niw_sweep_direct_kernel keeps live: inside
// that kernel every variant spilled 100-140 vector registers and ran 30-50 % slower (docs/experiments/r05_bf16x3_sublabels.patch).  Here the
// evaluation lives in kernels that carry nothing else:
//   niw_lean_kernel   a whole tile in one go where the screens settle it: every point of the wave had label k0, the certified bf16 bracket
//                     of a_k0 + the ball test + the 4-row tail screens exclude every other cluster for every point -- then z = k0 by the
//                     same proof as in niw_sweep_direct_kernel (its `ref_skipped` draw) and only the sub-labels need values.  One
//                     conversion of z = x - mu_k0 into planes serves the bracket (plane h) and both sub-cluster evaluations.  Tiles it
//                     cannot settle go to a list and are left untouched.
//   niw_sub_kernel    the sub-label phase alone for the tiles of that list (or all tiles), behind niw_sweep_direct_kernel<.., LSTORE>
//                     which draws their labels and stores them.
// INVARIANT kept: a sub-cluster value is a function of (point, matrix) alone -- z is taken relative to the mean of the point's OWN (new)
// cluster, whatever tile or kernel the point is in -- so shards, tile schedules, debug tables and the two kernels give the same bits.
//
// The arithmetic.  Float32 values split EXACTLY into three bf16 planes, v = h + m + l (bf16x3_plane_bits: 3 x 8 significand bits); bf16 x bf16
// products are exact in the Float32 accumulator; of the nine plane products the six with weight >= 2^-16 are kept,
//     y = Rh zh + (Rh zm + Rm zh) + (Rh zl + Rm zm + Rl zh),
// the three dropped ones are below 2^-24 |R||z| per term -- the size of ONE Float32 rounding of the product, which the Float32 chain commits
// on every term.  z = x - mu_k (one Float32 subtraction per feature, as the reference's x - mu), converted once per label and shared by the
// left and the right evaluation:  R_s (x - mu_s) = R_s z + d_s,  d_s = R_s (mu_k - mu_s)  (niw_b3_pack_kernel: Float64 sums, rounded once;
// |mu_k - mu_s| is a sub-cluster's offset inside its own cluster, d_s is of the size of y itself: nothing cancels), d_s the accumulator's
// initial value.  Per matrix and 64 points: 144 bf16 matrix instructions of 16 cycles instead of 164 Float32 ones of 32.
// Image of a matrix: planes h | m | l, each the six fragments of refb_map ([64 lanes][4 dwords]): B3_WORDS dwords, behind the bracket's
// images in the `tail` buffer (b3_images / b3_offsets).
#include "dpmm_device.h"
#include "dpmm_kernels.h"
#include "niw_device.h"
#include "niw_b3.h"

namespace dpmm {

// ------------------------------------------------------------------------------------------------------------------ images
// Three-plane bf16 images and offset vectors of the 2K sub-cluster factors from the Float32 fragment image both pack kernels write
// (NB = 4: Rp [3K][10][64][4], mup [3K][64]); one workgroup per sub-cluster matrix.
__global__ __launch_bounds__(256) void niw_b3_pack_kernel(const float *__restrict__ Rp, const float *__restrict__ mup, float *__restrict__ tail, int K, int y0) {
    constexpr int NP = 10;
    uint32_t *img = const_cast<uint32_t *>(b3_images(tail, K));
    float *dvec = const_cast<float *>(b3_offsets(tail, K));
    const int k = blockIdx.x >> 1, j = 3 * k + 1 + (blockIdx.x & 1);
    const float *Rj = Rp + (size_t)j * NP * 256;
    auto elem = [&](int row, int col) -> float {            // R[row][col] out of the fragment image (0 below the diagonal)
        const int bi = row >> 4, t = col >> 4;
        if (t < bi) return 0.f;
        const int pair = pair_base<4>(bi) + (t - bi);
        const int ln = (row & 15) + 16 * ((col & 15) >> 2);
        return Rj[(size_t)pair * 256 + ln * 4 + (col & 3)];
    };
    const int yy = (int)blockIdx.y + y0;      // 0 .. 2: planes, 3: offsets, 4: the pair-ball table
    if (yy == 4) {                      // the table's column of cluster k (one workgroup per CLUSTER: the even ones of this row)
        if (blockIdx.x & 1) return;
        const float *Rk = Rp + (size_t)(3 * k) * NP * 256;
        auto elemR = [&](int row, int col) -> float {
            const int bi = row >> 4, t = col >> 4;
            return Rk[(size_t)(pair_base<4>(bi) + (t - bi)) * 256 + ((row & 15) + 16 * ((col & 15) >> 2)) * 4 + (col & 3)];
        };
        pair_ball_block<false>(elemR, [&](int kk, int c) -> float { return mup[(size_t)(3 * kk) * 64 + c]; }, K, k, const_cast<float *>(pair_ball_table(tail, K)));
        return;
    }
    if (yy < 3) {                       // one plane of the matrix: 1536 dwords, six per thread
        const int plane = yy;
        uint32_t *out = img + (size_t)j * B3_WORDS + (size_t)plane * REFB_WORDS;
        for (int e = threadIdx.x; e < REFB_WORDS; e += 256) {
            int row, c0, c1;
            refb_map(e, row, c0, c1);
            out[e] = bf16x3_plane_bits(elem(row, c0), plane) | (bf16x3_plane_bits(elem(row, c1), plane) << 16);
        }
    } else {                            // the offsets d = R_s (mu_k - mu_s): four threads per row, 16 columns each, Float64
        const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
        double acc = 0.0;
#pragma unroll 4
        for (int c = 16 * part; c < 16 * part + 16; ++c)
            if (c >= row) acc += (double)elem(row, c) * ((double)mup[(size_t)(3 * k) * 64 + c] - (double)mup[(size_t)j * 64 + c]);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) dvec[(size_t)j * B3_DVEC + row] = (float)acc;
    }
}
hipError_t launch_niw_b3_pack(const float *Rp, const float *mup, int K, float *tail, int what, hipStream_t s) {      // what bit 0: images + offsets, bit 1: the pair-ball table
    const bool pb = (what & 2) && K >= 2 && K <= PB_MAXK;
    if (K < 1 || !((what & 1) || pb)) return hipSuccess;
    const int y0 = (what & 1) ? 0 : 4, ny = (pb ? 5 : 4) - y0;
    DPMM_LAUNCH(niw_b3_pack_kernel, dim3(2 * K, ny), dim3(256), 0, s, Rp, mup, tail, K, y0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------ the pair-ball table
// A ball test in ALL D features on quantities that do not depend on the tile (round 6).  For every x, with c = mu_k0 the mean of the tile's
// reference cluster:   |R_j (x - mu_j)| >= |R_j (c - mu_j)| - |R_j (x - c)| >= D_k0,j - |R_j|_2 |x - c|,
// so  a_j(x) <= cst_j - 1/2 max(0, D_k0,j - s_j r)^2  for every point of a tile whose points lie within r of c: ONE comparison per cluster and
// tile (lane = cluster) once D_k0,j = |R_j (mu_k0 - mu_j)| and s_j >= |R_j|_2 are tabulated per parameter set.  The 4-feature ball test that
// follows it (ball_far_rec) sees a distance of ~ sqrt(4) standard deviations of the means where this one sees sqrt(D): on the bench data it leaves
// 3.1 tail-pair tests and 0.8 bottom screens per tile (23 and 5.9 at K = 256), this one none.
// One workgroup per cluster j (pair_ball_block, niw_b3.h: a row of workgroups of niw_b3_pack_kernel, or a role of the device master's hand-over):  s_j^2 <= max row sum of |R_j' R_j| (an upper bound of its largest eigenvalue; within ~1.4 of it for the factors of
// Wishart-like precisions), in Float32 with the slack of its own rounding;  D_k,j for every k from the Float32 means the sweep subtracts
// (wave w takes k = w, w + 4, ..: lane = row of R_j, the 64 differences broadcast by v_readlane).  Slack: the matrix-vector product in Float32
// errs by <= 64 * 2^-24 * |R_j|_F |v| <= 4e-5 s_j |v|, taken off D; D itself is rounded DOWN by 1e-4.
// Diagnostic (dpmm_debug_subloglik while the bf16 evaluation is active): for every point of the shard and every cluster k the two values a
// sweep would compute if the point's label were k -- out[(2k + s) n + i] -- through the same device functions, one wave per 64 points
__global__ __launch_bounds__(256) void niw_b3_debug_kernel(NiwSweepArgs A, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, ci = lane & 15, g = lane >> 4;
    const int64_t wbase = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (wbase >= A.n) return;
    const int myp32 = wbase + lane < A.n ? (int)(wbase + lane) : -1;
    for (int k = 0; k < A.K; ++k) {
        f32x4 x[4][4], mk[4];
        gather_x64(A.X, A.ldx, myp32, ci, g, x);
        b3_mean(A.mup, k, g, mk);
        B3Z Z;
        b3_convert(x, mk, Z);
        float bl, br;
        const B3Head H = b3_head(A.tail, A.cst, A.K, k, lane, g);
        b3_eval(A.tail, A.K, k, Z, H, lane, g, bl, br);
        if (myp32 >= 0) { out[(int64_t)(2 * k) * A.n + myp32] = bl; out[(int64_t)(2 * k + 1) * A.n + myp32] = br; }
    }
}
hipError_t launch_niw_b3_debug(const NiwSweepArgs &a, float *out, hipStream_t s) {
    if (!a.tail || a.n <= 0) return hipErrorInvalidValue;
    DPMM_LAUNCH(niw_b3_debug_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a, out);
    return hipGetLastError();
}

// (Rounds 5-6 pulled the NEXT tile's x rows towards L2 from here -- one dword per 128-byte line behind the evaluations' last fragment request, loaded by
// inline asm into two registers kept away from the allocator -- for 2 % of the launch when it was built.  Measured again at the end of round 6, after
// the records moved to LDS and the uniforms under the gather: with and without it the launch takes the same time (N = 1e7: 1.011 / 1.024 ms without,
// 1.020 / 1.019 with; 8-GPU shard 0.146 / 0.147; overlapping clusters 1.199 / 1.204 and 1.483 / 1.466), and the lines it pulled were fetched twice where the
// gather missed them (2.83 GB of traffic against 2.60 algorithmic).  Removed, with the reserved registers and the build-time check they needed.)

// ------------------------------------------------------------------------------------------------------------------ sub-labels alone
// The sub-label phase of the tiles named in `list` (list[0] = their number, list[1 ..] = wave-tile indices; null: every tile): the new labels are
// in bins (niw_sweep_direct_kernel<.., LSTORE> stored 2 z + old sub-label), the second uniform of the point's Philox draw decides between left
// and right (create_subclusters_labels!, local_clusters_actions.jl:83-95).  One wave per tile of 64 positions of the visiting order.
// (LISTED: with a list.  Two instantiations: in ONE kernel the words of a span -- loaded with a list, computed without -- shared registers, and the
//  no-list path waited for every outstanding load before it wrote them)
template <bool LISTED>
__global__ __launch_bounds__(256, 2) void niw_sub_kernel(NiwSweepArgs A, const uint32_t *__restrict__ list, uint32_t *__restrict__ count_out) {
    // (the list's length for the host's regime decision, written to pinned memory by the last launch that reads it: no copy launch)
    if (LISTED && count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = list[0];
    const int lane = threadIdx.x & 63, ci = lane & 15, g = lane >> 4;
    const bool use_order = A.order != nullptr && *A.order_total == (int32_t)A.n;
    const int64_t nwtiles = (A.n + 63) / 64;
    const int64_t count = LISTED ? (int64_t)list[0] : nwtiles;
    const int wave_id = (int)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int)gridDim.x * 4;
    unsigned nw_b3 = 0;
    // A tile's chain list -> order -> bins -> x is four dependent round trips; the first three are taken off it: the tile index is known two
    // tiles ahead, the point indices one tile ahead (requested at a tile's top), their labels requested once the indices have arrived (behind
    // this tile's x gather) -- as niw_lean_kernel does.  All-tiles mode without them: 0.82 ms at N = 1e7.
    // entry i of the list = (first position, number of positions <= 64): the lean kernel's tiles are aligned to the sort's bins; without a
    // list tile i is positions 64 i ..  A span is packed as position << 7 | count (n < 2^31, count <= 64); -1: none.
    // (What a tile's top needs of loaded values -- the span two tiles ahead, the labels of this tile's points -- is FORMED one tile earlier, at the end
    //  of the tile that requested it: formed at the top, behind the index load issued there, every tile began with s_waitcnt vmcnt(0).)
    auto span_words = [&](int64_t i, uint32_t &w0, uint32_t &w1) {           // the two words of span i as loaded (combined by pack_span when they have landed)
        w0 = 0u; w1 = 0xffffffffu;                                          // (w1 = ~0: none)
        if (i >= count) return;
        if constexpr (LISTED) { w0 = list[1 + 2 * i]; w1 = list[2 + 2 * i]; }
        else { const int64_t e = A.n - 64 * i; w0 = (uint32_t)(64 * i); w1 = (uint32_t)(e < 64 ? e : 64); }
    };
    auto pack_span = [](uint32_t w0, uint32_t w1) -> int64_t { return w1 == 0xffffffffu ? (int64_t)-1 : (((int64_t)w0 << 7) | (int64_t)w1); };
    auto label_of = [&](int bin) -> int { const int zz = bin >> 1; return (unsigned)zz < (unsigned)A.K ? zz : -1; };      // (a label outside [0, K): left alone)
    uint32_t sa, sb, sa2, sb2;
    span_words(wave_id, sa, sb);
    span_words((int64_t)wave_id + nwaves, sa2, sb2);
    int64_t t_next = pack_span(sa, sb);
    int nx_p = -1, nx_z = -1;
    if (t_next >= 0) {
        const int64_t pos = (t_next >> 7) + lane;
        if (lane < (int)(t_next & 127)) { nx_p = use_order ? A.order[pos] : (int)pos; nx_z = label_of(A.bins[nx_p]); }
    }
    for (int64_t idx = wave_id; idx < count; idx += nwaves) {
        t_next = pack_span(sa2, sb2);
        const int myp32 = nx_p;
        const int z = nx_z;
        int pf_p = -1, pf_bin = -1;
        const bool has_next = t_next >= 0 && lane < (int)(t_next & 127);      // (the lane has a point in the next tile: known without its index)
        bool bin_asked = false;                                                // (wave-uniform)
        if (has_next) {
            const int64_t posn = (t_next >> 7) + lane;
            pf_p = use_order ? A.order[posn] : (int)posn;
        }
        span_words(idx + 2 * (int64_t)nwaves, sa2, sb2);                      // (requested behind the last use of the words it replaces)
        float u_sub = 0.f;
        if (z >= 0) u_sub = u01(philox4x32_10(A.seed, (uint64_t)(A.first_index + myp32), A.epoch, STREAM_SWEEP).v[1]);
        float b0 = -INFINITY, b1 = -INFINITY;
        unsigned long long todo = __ballot(z >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int k = __builtin_amdgcn_readfirstlane(__shfl(z, leader));
            todo &= ~__ballot(z == k);
            f32x4 x[4][4], mk[4];
            gather_x64(A.X, A.ldx, myp32, ci, g, x);
            b3_mean(A.mup, k, g, mk);
            const B3Head H = b3_head(A.tail, A.cst, A.K, k, lane, g);       // (requested with x: the conversion covers their round trip)
            __builtin_amdgcn_sched_barrier(0);
            B3Z Z;
            b3_convert(x, mk, Z);
            if (!bin_asked) {                                                 // the next tile's labels (its indices arrived with x)
                int q = pf_p;
                asm volatile("" : "+v"(q));                                   // (the address is formed HERE: hoisted out of the loop it waited for the index in front of the gather)
                if (has_next) pf_bin = A.bins[q];
                bin_asked = true;
            }
            float bl, br;
            b3_eval(A.tail, A.K, k, Z, H, lane, g, bl, br);
            if (z == k) { b0 = bl; b1 = br; }
            nw_b3 += 2;
        }
        if (z >= 0) A.bins[myp32] = 2 * z + draw2(b0, b1, u_sub);
        if (!bin_asked && has_next) pf_bin = A.bins[pf_p];                   // (a tile without a label in range)
        nx_p = pf_p; nx_z = has_next ? label_of(pf_bin) : -1;
        asm volatile("" : "+v"(nx_z));                                          // (here, not sunk to the next tile's top)
    }
    if (A.work && lane == 0) A.work[DPMM_WORK_SLOTS + (size_t)wave_id * DPMM_WORK_PER_WAVE + 7] += (unsigned long long)nw_b3 << 32;
}
hipError_t launch_niw_sub(const NiwSweepArgs &a, const uint32_t *list, uint32_t *count_out, int grid, hipStream_t s) {
    if (!a.tail || a.n <= 0) return hipErrorInvalidValue;
    if (list) DPMM_LAUNCH(niw_sub_kernel<true>, dim3(grid), dim3(256), 0, s, a, list, count_out);
    else DPMM_LAUNCH(niw_sub_kernel<false>, dim3(grid), dim3(256), 0, s, a, list, count_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------ the lean kernel
// The reference bracket of niw_sweep_direct_kernel (ref_bracket, niw_sweep.hip) on operands that already exist: plane h of z = x - mu_k0 IS
// the bracket's bf16 operand (same subtraction, same v_cvt_pk_bf16_f32), so q_hi comes out bit for bit as there.
__device__ __forceinline__ void ref_bracket_planes(const u32x4_t (&a)[6], const B3Z &Z, float (&qhi)[4]) {      // a: the six fragments of k0's bracket image (this lane's)
    const u32x4_t absm = (u32x4_t){0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu};
#pragma unroll
    for (int n0 = 0; n0 < 4; n0 += 2) {
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
                    const u32x4_t zb = Z.p[n0 + h][sl][0];
                    y[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[f]), __builtin_bit_cast(bf16x8_t, zb), y[h], 0, 0, 0);
                    e[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aa), __builtin_bit_cast(bf16x8_t, zb & absm), e[h], 0, 0, 0);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = __builtin_fmaf(REFB_C, e[h][r], fabsf(y[h][r]));
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

// A whole tile (64 positions of the visiting order) in one go WHERE THE SCREENS SETTLE IT: every point of the wave had label k0; the certified
// bracket of a_k0 (lower end), the ball test and the 4-row tail screens -- the first stages of niw_sweep_direct_kernel's cascade, same
// records, same thresholds -- exclude every other cluster for every point.  Then that kernel's draw returns k0 whatever a_k0 is (its
// `ref_skipped` branch; uniforms are in (0, 1): u01, dpmm_device.h), and what remains is the sub-label phase on planes
// that exist already.  Every other tile (mixed previous labels, a candidate left, no previous labels) is appended to `list` ([0] = count,
// cleared before the launch) and left untouched: niw_sweep_direct_kernel<.., LSTORE> + niw_sub_kernel take it.  The outcome is that of the
// one-kernel path: a tile is settled here only if every later screen there would find nothing to do either.
#ifdef DPMM_STAMPS
#define LSTAMP(var) unsigned long long var; do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LSTAMP(var)
#endif
template <bool DIR>
__device__ __forceinline__ void niw_lean_body(NiwSweepArgs A, uint32_t *__restrict__ list, uint32_t *__restrict__ need2, uint32_t *__restrict__ other_list,
                                              const int32_t *__restrict__ bin_start, int nbins, uint32_t *__restrict__ need3) {
    // two lists take turns: this launch appends to `list` (count cleared by the previous lean launch) and clears the other one's count for the
    // next -- every reader of that one finished before this launch started (stream order).  No fill launch in front of a sweep.
    if (other_list && blockIdx.x == 0 && threadIdx.x == 0) other_list[0] = 0u;
#ifdef DPMM_STAMPS
    unsigned long long T_x = 0, T_conv = 0, T_br = 0, T_scr = 0, T_u = 0, T_p2 = 0, T_tot = 0; int ntl = 0;
#endif
    const int lane = threadIdx.x & 63, ci = lane & 15, g = lane >> 4;
    const bool use_order = A.order != nullptr && *A.order_total == (int32_t)A.n;
    const int K = A.K;
    const int wave_id = (int)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int)gridDim.x * 4;
    unsigned nw_easy = 0, nw_br = 0, nw_tail = 0, nw_bb = 0;
    // THE DIRECTION SCREEN IN THIS KERNEL (round 6).  While the library keeps the screen's tables (overlapping clusters: tiles keep eight or more
    // candidates behind the 4-row tests; A.sp_frag / A.sp_cons, K <= 64) rounds 4-5 ran no lean launch at all: nothing settles without the screen,
    // and the two launches that took every tile instead cost 1.9 ms at N = 1e7 against 1.0.  The screen's operand is plane h of z0 = x - mu_k0 --
    // it exists here -- and |z0| is accumulated by the conversion: a tile that keeps six or more candidates behind the ball test puts all of them
    // through direction_far_core (niw_device.h: the general kernel's own matrix product and bounds); what is left takes the 4-row pairs and the
    // bf16 bottom screens as before.  Every exclusion is a certified bound, so a tile settled here is one whose draw returns k0.
    // The statistics the library's regime switch reads (candidates per tile, the screen's yield) go to need3 in the general kernel's format,
    // for the tiles SETTLED here (a tile handed on is counted by the launch that finishes it).
    // (an instantiation of its own, as in the general kernel: the screen's code in the ONE kernel cost the tiles that never use it 8 spilled vector
    //  registers and 10 % -- lean launch 1.00 -> 1.11 ms at N = 1e7 on the bench data)
    const bool use_dir = DIR && A.sp_frag != nullptr && A.sp_cons != nullptr && A.bf16scr && K >= 3 && K <= SP_MAXK;
    const bool dir_first = use_dir && (A.bf16scr & 2) != 0;
    unsigned nw_sp = 0, nw_cand = 0, nw_dcand = 0, nw_dexcl = 0, nw_ctiles = 0;
    // the ball test's records (16 floats per cluster, the same for every tile) once per workgroup in LDS: the test then waits for an LDS read
    // instead of an L2 round trip per tile
    constexpr int BALL_LDS_K = 256;
    __shared__ __attribute__((aligned(16))) float ball_lds[BALL_LDS_K * 16];
    const bool ball_in_lds = A.ball && K <= BALL_LDS_K;
    // the pair-ball test (A.ball bit 1: the table of niw_pair_ball_kernel belongs to these parameters; not in the direction-screen instantiation: on
    // overlapping clusters it clears nothing): the norm bounds s_j once per workgroup in LDS, the row of distances of a tile's k0 requested per tile
    __shared__ float pb_s[PB_MAXK];
    const bool use_pb = !DIR && (A.ball & 2) != 0 && ball_in_lds && K <= PB_MAXK;
    const float *pb_d = pair_ball_table(A.tail, K);
    if (use_pb)
        for (int e = threadIdx.x; e < K; e += 256) pb_s[e] = pb_d[(size_t)K * K + e];
    if (ball_in_lds) {
        const float *src = ball_records(A.tail, K);
        for (int e = threadIdx.x; e < 16 * K; e += 256) ball_lds[e] = src[e];
    }
    // TILES ALIGNED TO THE SORT'S BINS.  The visiting order is sorted by bin (label, sub-label); a tile of 64 consecutive positions that
    // crosses from one cluster into the next has mixed previous labels and goes to the general path -- 31 such tiles per sweep at K = 32,
    // whatever N, and the two launches that take them are a tile's latency each (45 us of the 8-GPU shard's 355).  With bin_start (the sort's
    // bin offsets, nbins + 1 entries) a tile never crosses a bin: bin b owns ceil(count_b / 64) tiles, its last one partly filled.
    // Whole-bin relabels since the sort (split, merge) keep a tile's points in one cluster; a tile is still CHECKED against the labels.
    constexpr int MAXB = NIW_LEAN_MAX_BINS;
    __shared__ int tstart_s[MAXB + 2], bstart_s[MAXB + 2], wsum_s[4];
    const bool aligned = bin_start != nullptr && use_order && nbins >= 1 && nbins <= MAXB && bin_start[nbins] == (int32_t)A.n;      // (workgroup-uniform)
    const int nb = aligned ? nbins : 1;                     // (no usable table: ONE bin holding every position -- tiles of 64 consecutive positions)
    {
        const int tid = threadIdx.x;
        int c = 0;
        if (tid < nb) {
            const int s0 = aligned ? bin_start[tid] : 0, s1 = aligned ? bin_start[tid + 1] : (int)A.n;
            c = (s1 - s0 + 63) >> 6; bstart_s[tid] = s0;
        }
        if (tid == 0) bstart_s[nb] = (int)A.n;                    // (by thread 0, not by thread nb: nb may equal the workgroup size -- ADVICE r5)
        int inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
        if (lane == 63) wsum_s[tid >> 6] = inc;
        __syncthreads();
        const int w = tid >> 6;
        inc += (w > 0 ? wsum_s[0] : 0) + (w > 1 ? wsum_s[1] : 0) + (w > 2 ? wsum_s[2] : 0);
        tstart_s[tid + 1] = inc;
        if (tid == 0) tstart_s[0] = 0;
    }
    __syncthreads();
    const int ntiles_all = __builtin_amdgcn_readfirstlane(tstart_s[nb]);
    int bptr = 0;
    // the span of tile t (first position, number of positions); the tiles a wave asks for increase: one running bin pointer
    auto span_of = [&](int t, int &p0, int &cn) {
        p0 = 0; cn = 0;
        if (t >= ntiles_all) return;
        while (t >= __builtin_amdgcn_readfirstlane(tstart_s[bptr + 1])) ++bptr;
        p0 = __builtin_amdgcn_readfirstlane(bstart_s[bptr]) + 64 * (t - __builtin_amdgcn_readfirstlane(tstart_s[bptr]));
        const int e = __builtin_amdgcn_readfirstlane(bstart_s[bptr + 1]) - p0;
        cn = e < 64 ? e : 64;
    };
    // Off a tile's critical chain order -> bins -> x (three dependent round trips): the point indices and previous labels of the next tile are
    // fetched while this one is processed.
    auto index_at = [&](int p0, int cn) -> int { return lane < cn ? (use_order ? A.order[p0 + lane] : p0 + lane) : -1; };
    int c_p0, c_cn, n_p0, n_cn;                              // spans of this tile and of the next one
    span_of(wave_id, c_p0, c_cn);
    span_of(wave_id + nwaves, n_p0, n_cn);
    int nx_p = index_at(c_p0, c_cn);
    // carried from tile to tile: the previous LABEL of the lane's point (-1: none), formed at the END of the tile that fetched the bin word, where that
    // load has long landed.  (Formed at the next tile's top it sat behind the index load issued there and the compiler answered with s_waitcnt
    // vmcnt(0) -- that load, the last store's acknowledgement -- in front of every gather.  Measured: no change of the launch time; kept for the shorter chain.)
    auto label_of = [&](int bin) -> int { const int pv = bin >= 0 ? (bin >> 1) : -1; return (unsigned)pv < (unsigned)K ? pv : -1; };
    int nx_prev = label_of(nx_p >= 0 ? A.bins[nx_p] : -1);
    for (int tile = wave_id; tile < ntiles_all; tile += nwaves) {
        const bool valid = lane < c_cn;
        const int myp32 = nx_p;
        LSTAMP(s0);
#ifdef DPMM_STAMPS
        unsigned long long s1 = s0, s2 = s0, s3 = s0, s4 = s0, s5 = s0;
#endif
        const int pf_p = index_at(n_p0, n_cn);                   // the next tile's indices
        int pf_bin = -1;                                         // its previous labels: requested when the indices have arrived (with x)
        bool bin_asked = false;                                  // (wave-uniform)
        const int prev = nx_prev;
        const unsigned long long pm = __ballot(prev >= 0);
        bool hard = pm == 0ull;
        int k0 = 0;
        if (!hard) {
            k0 = __builtin_amdgcn_readfirstlane(__shfl(prev, __ffsll((long long)pm) - 1));
            hard = __ballot(valid && prev != k0) != 0ull;
        }
        B3Z Z;
        B3Head H;
        float u_sub = 0.f;
        if (!hard) {
            f32x4 xt = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 x3[4];                                           // the last 16 features (the bf16 bottom screens' operand)
            float nz[4] = {0.f, 0.f, 0.f, 0.f};                   // |z0| per point group's point (direction screen)
            u32x4_t abr[6];                                        // k0's bracket fragments: requested with x, used behind the conversion
            {
                f32x4 x[4][4], mk[4];
                gather_x64(A.X, A.ldx, myp32, ci, g, x);
                b3_mean(A.mup, k0, g, mk);
                {
                    const u32x4_t *F = reinterpret_cast<const u32x4_t *>(refb_records(A.tail, K) + (size_t)k0 * REFB_WORDS) + (unsigned)lane;
#pragma unroll
                    for (int f = 0; f < 6; ++f) abr[f] = F[64 * f];
                }
                __builtin_amdgcn_sched_barrier(0);
                // the point's uniforms need nothing of x: their ~150 vector instructions run while the gather is on its way
                if (valid) {
                    const Philox4 r = philox4x32_10(A.seed, (uint64_t)(A.first_index + myp32), A.epoch, STREAM_SWEEP);
                    u_sub = u01(r.v[1]);               // (the label's uniform r.v[0] is in (0, 1) -- u01, dpmm_device.h -- and the draw of a settled tile returns k0 whatever it is)
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef DPMM_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                { LSTAMP(t1); s1 = t1; }
#endif
                // the last four features of "this lane's point" (the tail screens' operand), as niw_sweep_direct_kernel takes them
                const int src = ci + 16 * A.tail_g;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    f32x4 v;
                    v.x = __shfl(x[n][3].x, src); v.y = __shfl(x[n][3].y, src); v.z = __shfl(x[n][3].z, src); v.w = __shfl(x[n][3].w, src);
                    if (g == n) xt = v;
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) x3[n] = x[n][3];
                if constexpr (DIR) {
                    if (use_dir) {
                        float part[4];
                        b3_convert<true>(x, mk, Z, part);       // x's last use (but for x3)
#pragma unroll
                        for (int n = 0; n < 4; ++n) nz[n] = direction_norm(part[n]);
                    } else b3_convert(x, mk, Z);
                } else b3_convert(x, mk, Z);                   // x's last use (but for x3)
            }
            if (lane < n_cn) pf_bin = A.bins[pf_p];            // the next tile's previous labels (its indices have arrived with x)
            bin_asked = true;
            // pair-ball distance D_k0,j of "this lane's cluster" j = lane (requested here: the bracket covers the trip; clusters 64 .. are
            // requested in their turn -- four registers across the bracket were three spilled ones)
            float pbv0 = 0.f;
            if (use_pb) pbv0 = pb_d[(size_t)k0 * K + (lane < K ? lane : 0)];
#ifdef DPMM_STAMPS
            { LSTAMP(t2); s2 = t2; }
#endif
            float qhi[4];
            ref_bracket_planes(abr, Z, qhi);
            ++nw_br;
#ifdef DPMM_STAMPS
            { LSTAMP(t3); s3 = t3; }
#endif
            const float c0 = A.cst[3 * k0];
            float my_best = -INFINITY;
            float thrb[4];                                         // per point group: threshold of the column's point (+inf for a column without a point)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float bn = __builtin_fmaf(-0.5f, qhi[n], c0);
                if (g == n) my_best = bn;
                thrb[n] = (16 * n + ci < c_cn) ? bn - A.screen_margin : INFINITY;
            }
            const float my_thr = valid ? my_best - A.screen_margin : INFINITY;
            // the 4-feature ball of the wave is formed only for a tile the pair-ball test leaves a candidate of (its centre is a scalar load away)
            BallWave ball; ball.ok = false;
            bool ball_made = false;
            float thr_min = INFINITY;
            const bool thr_ok = use_pb && __ballot(valid && !(my_thr == my_thr)) == 0ull;       // (the pair-ball test's threshold: the wave's lowest)
            if (use_pb) thr_min = wave_min_f32(my_thr);
            else if (A.ball) { ball = ball_of_wave(A.tail, K, k0, xt, my_thr, valid); ball_made = true; }
            // the pair-ball radius: r >= max |x - mu_k0| over the tile's points.  |z_h|^2 of 16 points at a time is the diagonal of the Gram matrix of
            // plane h with itself (the lane's registers ARE both operands: row = column = lane & 15, the same eight features per lane group);
            // |z| <= |z_h| / (1 - 2^-9) component by component, the products are exact, their Float32 sum of 64 non-negative terms errs by < 4e-6:
            // the factor 1.005 covers both.  Point p of a group sits in lane group p / 4, register p % 4.
            float pb_r = INFINITY;
            if (thr_ok) {
                float m2 = 0.f;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    f32x4 gq = (f32x4){0.f, 0.f, 0.f, 0.f};
                    gq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, Z.p[n][0][0]), __builtin_bit_cast(bf16x8_t, Z.p[n][0][0]), gq, 0, 0, 0);
                    gq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, Z.p[n][1][0]), __builtin_bit_cast(bf16x8_t, Z.p[n][1][0]), gq, 0, 0, 0);
                    const float dsel = (ci & 2) ? ((ci & 1) ? gq[3] : gq[2]) : ((ci & 1) ? gq[1] : gq[0]);
                    const bool mine = g == (ci >> 2) && 16 * n + ci < c_cn;
                    m2 = fmaxf(m2, mine ? dsel : 0.f);
                }
                const float mx = wave_max_f32(m2);
                pb_r = (mx == mx) ? __builtin_amdgcn_sqrtf(mx) * 1.005f : INFINITY;
            }
            for (int base = 0; base < K && !hard; base += 64) {
                unsigned long long cand = (K - base >= 64) ? ~0ull : ((1ull << (K - base)) - 1ull);
                if (k0 >= base && k0 < base + 64) cand &= ~(1ull << (k0 - base));
                if (pb_r < INFINITY) {                             // (wave-uniform)
                    const int j = base + lane, jj = j < K ? j : 0;
                    const float dv = base == 0 ? pbv0 : pb_d[(size_t)k0 * K + jj];
                    const float lb = fmaxf(__builtin_fmaf(-pb_s[jj], pb_r, dv), 0.f);
                    const float ub = __builtin_fmaf(-0.5f * lb, lb, ball_lds[16 * jj + 15]);
                    cand &= ~__ballot(j < K && ub < thr_min);
                }
                if (cand && A.ball && !ball_made) { ball = ball_of_wave(A.tail, K, k0, xt, my_thr, valid); ball_made = true; }
                if (ball.ok && cand) {
                    const int j = base + lane;
                    cand &= ~(ball_in_lds ? ball_far_rec(ball_lds + 16 * (j < K ? j : 0), j < K, ball) : ball_far(A.tail, K, base, lane, ball));
                }
                unsigned tile_cand = 0, tile_dc = 0, tile_dx = 0, tile_sp = 0;       // (this tile's share of the statistics: counted only if it is settled here)
                auto direction = [&]() {
                    const int nc = __builtin_popcountll(cand);
                    if (nc >= 6) {                                    // (below: the candidates' own screens are cheaper)
                        u32x4_t zb[4][2];
#pragma unroll
                        for (int n = 0; n < 4; ++n) { zb[n][0] = Z.p[n][0][0]; zb[n][1] = Z.p[n][1][0]; }
                        cand &= ~direction_far_core<4>(A.sp_frag + (size_t)k0 * SP_FRAG_WORDS, A.sp_cons + (size_t)k0 * SP_CONS_FLOATS, zb, nz, thrb, lane, g, K);
                        tile_sp = 1; tile_dc = (unsigned)nc; tile_dx = (unsigned)(nc - __builtin_popcountll(cand));
                    }
                };
                if constexpr (DIR) { if (dir_first) { tile_cand = (unsigned)__builtin_popcountll(cand); direction(); } }
                for (unsigned long long pend = cand; pend;) {
                    const int sh = __builtin_ctzll(pend) & ~1;            // (base is a multiple of 64: pair 2p sits at an even bit)
                    const int pr = (base + sh) >> 1;
                    pend &= ~(3ull << sh);
                    ++nw_tail;
                    cand &= ~((unsigned long long)tail_pair_far(tail_load_pair(A.tail, pr), xt, my_thr) << sh);
                }
                if constexpr (DIR) { if (use_dir && !dir_first) { tile_cand = (unsigned)__builtin_popcountll(cand); direction(); } }       // (a measuring sweep: what the 4-row tests leave)
                // what the 4-row tests leave (0.8 clusters per tile on the bench data): the bf16 bottom screen, as niw_sweep_direct_kernel runs it
                // next (fragment 5 of the candidate's bracket image, its last 16 means, its constant); a candidate that passes it is left to that kernel
                if (A.bf16scr) {
                    const u32x4_t *Rb0 = reinterpret_cast<const u32x4_t *>(refb_records(A.tail, K));
                    while (cand) {
                        const int k = base + __builtin_ctzll(cand);
                        const u32x4_t a5 = Rb0[(size_t)k * (REFB_WORDS / 4) + 64 * 5 + lane];
                        const f32x4 m4 = *reinterpret_cast<const f32x4 *>(A.mup + (size_t)(3 * k) * 64 + 48 + 4 * g);
                        ++nw_bb;
                        if (!bf16_bottom_excludes<4>(a5, x3, m4, A.cst[3 * k], thrb)) break;
                        cand &= cand - 1ull;
                    }
                }
                hard = cand != 0ull;
                if constexpr (DIR) { if (use_dir && !hard) { nw_cand += tile_cand; nw_sp += tile_sp; nw_dcand += tile_dc; nw_dexcl += tile_dx; ++nw_ctiles; } }
            }
        }
#ifdef DPMM_STAMPS
        { LSTAMP(t4); s4 = t4; }
#endif
        if (!hard) {
            H = b3_head(A.tail, A.cst, K, k0, lane, g);           // the sub-label evaluation's first fragments
        }
#ifdef DPMM_STAMPS
        { LSTAMP(t5); s5 = t5; }
#endif
        if (hard) {
            if (lane == 0) { const uint32_t at = atomicAdd(&list[0], 1u); list[1 + 2 * at] = (uint32_t)c_p0; list[2 + 2 * at] = (uint32_t)c_cn; }
            if (!bin_asked && lane < n_cn) pf_bin = A.bins[pf_p];      // (a tile that left before the bracket)
        } else {
            float bl, br;
            b3_eval(A.tail, K, k0, Z, H, lane, g, bl, br);
            if (valid) A.bins[myp32] = 2 * k0 + draw2(bl, br, u_sub);
            ++nw_easy;
        }
        nx_p = pf_p; nx_prev = label_of(pf_bin);
        asm volatile("" : "+v"(nx_prev));                      // (here, not sunk to the next tile's top)
        c_p0 = n_p0; c_cn = n_cn;
        span_of(tile + 2 * nwaves, n_p0, n_cn);
#ifdef DPMM_STAMPS
        { LSTAMP(s6); T_x += s1 - s0; T_conv += s2 - s1; T_br += s3 - s2; T_scr += s4 - s3; T_u += s5 - s4; T_p2 += s6 - s5; T_tot += s6 - s0; ++ntl; }
#endif
    }
#ifdef DPMM_STAMPS
    if (A.dbg && lane == 0) {
        unsigned long long *d = A.dbg + (size_t)(8192 + wave_id) * 16;
        d[0] = T_x; d[1] = T_conv; d[2] = T_br; d[3] = T_scr; d[4] = T_u; d[5] = T_p2; d[6] = 0; d[7] = T_tot; d[8] = ntl;
    }
#endif
    if (A.work && lane == 0) {
        unsigned long long *slot = A.work + DPMM_WORK_SLOTS + (size_t)wave_id * DPMM_WORK_PER_WAVE;      // (accumulates; cleared by the reader)
        slot[0] += nw_easy; slot[3] += nw_tail; slot[4] += nw_br; slot[5] += nw_bb; slot[7] += ((unsigned long long)(2 * nw_easy) << 32) + nw_sp;
    }
    // tiles settled here: without the screen they had no candidate behind the 4-row tests (need2: a plain count); with it, the candidates they had
    // and the screen's yield in the general kernel's two-word format (need3; bit 15: counted in front of the 4-row tests, an upper bound)
    if (need2 && lane == 0) need2[wave_id] = use_dir ? 0u : (nw_easy < 65535u ? nw_easy : 65535u);
    if (need3 && lane == 0) {
        uint32_t word = 0u, yield = 0u;
        if (use_dir) {
            word = ((nw_cand < 65535u ? nw_cand : 65535u) << 16) | (nw_ctiles < 32767u ? nw_ctiles : 32767u) | (dir_first ? 0x8000u : 0u);
            yield = ((nw_dexcl < 65535u ? nw_dexcl : 65535u) << 16) | (nw_dcand < 65535u ? nw_dcand : 65535u);
        }
        need3[2 * wave_id] = word; need3[2 * wave_id + 1] = yield;
    }
}
// niw_lean_kernel: the kernel of the bench data (no direction screen); niw_lean_kernel_dir: with the screen (a handful of spilled registers: loop-invariant
// addresses) -- launched while the library keeps the screen's tables
__global__ __launch_bounds__(256, 2) void niw_lean_kernel(NiwSweepArgs A, uint32_t *__restrict__ list, uint32_t *__restrict__ need2, uint32_t *__restrict__ other_list,
                                                         const int32_t *__restrict__ bin_start, int nbins, uint32_t *__restrict__ need3) {
    niw_lean_body<false>(A, list, need2, other_list, bin_start, nbins, need3);
}
__global__ __launch_bounds__(256, 2) void niw_lean_kernel_dir(NiwSweepArgs A, uint32_t *__restrict__ list, uint32_t *__restrict__ need2, uint32_t *__restrict__ other_list,
                                                             const int32_t *__restrict__ bin_start, int nbins, uint32_t *__restrict__ need3) {
    niw_lean_body<true>(A, list, need2, other_list, bin_start, nbins, need3);
}
hipError_t launch_niw_lean(const NiwSweepArgs &a, uint32_t *list, uint32_t *need2, uint32_t *other_list, const int32_t *bin_start, int nbins, uint32_t *need3, int grid, hipStream_t s) {
    if (!a.tail || a.n <= 0 || !list) return hipErrorInvalidValue;
    if (nbins > NIW_LEAN_MAX_BINS) { bin_start = nullptr; nbins = 0; }
    if (a.sp_frag && a.sp_cons && a.bf16scr && a.K >= 3 && a.K <= SP_MAXK) DPMM_LAUNCH(niw_lean_kernel_dir, dim3(grid), dim3(256), 0, s, a, list, need2, other_list, bin_start, nbins, need3);
    else DPMM_LAUNCH(niw_lean_kernel, dim3(grid), dim3(256), 0, s, a, list, need2, other_list, bin_start, nbins, need3);
    return hipGetLastError();
}

}  // namespace dpmm
