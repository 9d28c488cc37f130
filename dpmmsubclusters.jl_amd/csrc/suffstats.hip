// suffstats.hip -- per-(cluster, sub-cluster) sufficient statistics on gfx950.
//
// Stands in for create_suff_stats_dict_worker (src/local_clusters_actions.jl:149-169) with
// create_sufficient_statistics for the NIW prior (src/priors/niw.jl:42-51: N, sum x, X X' in
// Float64) and for the Multinomial prior (src/priors/multinomial_prior.jl:27-32: N, sum x).
//
// The reference builds a boolean mask per cluster and gathers; here the points are grouped
// once per pass by a stable counting sort on the bin = (label, sub-label) and every bin is
// then a contiguous segment of `perm`:
//   1. hist      one wave per SORT_TILE points, LDS counters -> tile_hist[bin][tile]
//   2. scan      one workgroup per bin: exclusive scan over tiles, bin totals
//   3. starts    bin_start (exclusive scan of totals) and the work-item table
//   4. scatter   stable rank inside the wave (ballot match loop) -> perm
//   5. stats     work item = (bin, <= chunk points): gather the columns, Float64 MFMA
//                (v_mfma_f64_16x16x4_f64) outer-product accumulation of the lower block
//                triangle + Float64 column sums, one slab per item
//   6. reduce    per bin: sum the item slabs in item order (deterministic), emit the packed
//                row {N, sum, lower triangle of S}
// Statistics are bitwise reproducible run to run (no floating-point atomics).
//
// f64 MFMA fragment conventions (lane l: i = l & 15, g = l >> 4): A[i][k=g], B[k=g][col=i],
// C/D element r (0..3): row g + 4r, col i.
#include <algorithm>
#include <cstdlib>
#include "dpmm_device.h"
#include "dpmm_kernels.h"

namespace dpmm {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------ sort
// `fast_total` (nullable): [nbins] running bin totals, added with integer atomics (order-independent, exact) -- the per-step pass needs the
// sub-cluster occupancies before anything is scanned (bad-cluster flags); starts_step_kernel clears them again for the next pass.
// `prev_lab` / `dirty` (nullable pair): the cluster label every point had at the previous per-step pass, and per-cluster flags "a point
// entered or left this cluster since then" -- what lets the statistics pass compute only the smaller sub-cluster of an untouched cluster
// and take the other one from the cached cluster-level row (derive_rows_kernel).  Any path that changes labels is seen here.
// W waves per workgroup, one sort tile each: the bin totals go to `fast_total` once per WORKGROUP.  With labels in no particular storage order
// every tile holds every bin, and one wave per workgroup meant tiles x bins atomics on `nbins` addresses (N = 1e6, K = 32: 125 k atomics, 2 k
// in a row per address at ~13 ns each = the kernel's 26 us; 33 us at N = 1e7).
// SPEC (round 6, DPMM_OPT_CHAIN_FUSION bit 8: no reset_recount launch).  reset_bad_clusters! re-draws the sub-labels of every cluster with an
// EMPTY sub-cluster -- known only when the whole shard is counted, which is why the reset was a launch of its own that read every label again
// (7 us at the 8-GPU shard size, 17-21 us at N = 1e7).  But a cluster that will be flagged has one sub-bin empty in EVERY tile, and the re-draw
// is a pure function of (seed, global point index, epoch): a tile can count the outcome ahead.  For every cluster that is one-sided IN THIS TILE
// (exactly one of its two sub-bins is empty here; ~never for a healthy cluster: 2^-64 at 64 of its points per tile) the wave evaluates the
// re-draw of those points, counts the result into `tile_spec` (written for such (tile, cluster) pairs only) and notes each point's new bin (`spec_bins`).  The scan
// then reads tile_spec for flagged clusters and tile_cnt for the rest, and the scatter stores spec_bins for the points of flagged clusters and places them by it
// (scatter_kernel<.., STEP>, RESET): same labels, same permutation as histogram -> reset_recount -> scan.
template <int TILE, int W, bool SPEC>
__global__ __launch_bounds__(64 * W) void hist_kernel(const int32_t *__restrict__ bins, int64_t n, int nbins, int nt,
                                                      int32_t *__restrict__ tile_hist, int32_t *__restrict__ fast_total,
                                                      uint16_t *__restrict__ prev_lab, uint8_t *__restrict__ dirty,
                                                      int32_t *__restrict__ tile_spec, int32_t *__restrict__ spec_bins, int64_t first, uint64_t seed, uint32_t epoch) {
    extern __shared__ int cnt_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *const cnt = cnt_all + wave * nbins;
    const int tile = blockIdx.x * W + wave;
    for (int b = lane; b < nbins; b += 64) cnt[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)tile * TILE;
    int4 v[TILE / 256];                      // (a full tile's labels stay in registers for the SPEC phase)
#pragma unroll
    for (int it = 0; it < TILE / 256; ++it) v[it] = make_int4(-1, -1, -1, -1);
    if (tile < nt) {
    auto track = [&](int64_t i, int bv) {            // rare: a point whose label is not the one it had at the previous pass
        // (a label outside [0, K) -- perm_total != n acknowledges that they can occur -- still LEAVES the cluster the point was in: that
        // cluster's cached row is stale whether or not the new bin is counted)
        const bool inr = (unsigned)bv < (unsigned)nbins;
        const unsigned z = inr ? (unsigned)bv >> 1 : 0xFFFFu, p = prev_lab[i];
        if (z != p) {
            if (inr) dirty[z] = 1;
            if (p < DPMM_MAX_CLUSTERS_K) dirty[p] = 1;
            prev_lab[i] = (uint16_t)z;
        }
    };
    if (base + TILE <= n) {
        // full tile: 16-byte loads, all of them in flight before the first is used; 256 points with one common bin (the usual case
        // after an ordered sweep: neighbours share a label) cost one LDS add instead of 256 same-address atomics
        const int4 *src = reinterpret_cast<const int4 *>(bins + base);
#pragma unroll
        for (int it = 0; it < TILE / 256; ++it) v[it] = src[it * 64 + lane];
        if (prev_lab) {
            // four 16-bit labels per lane and trip, all loads in flight with the bins'; ONE wave-level test for the whole tile (nothing
            // moved: the usual case) in front of the per-point bookkeeping
            const uint2 *psrc = reinterpret_cast<const uint2 *>(prev_lab + base);
            uint2 pv[TILE / 256];
#pragma unroll
            for (int it = 0; it < TILE / 256; ++it) pv[it] = psrc[it * 64 + lane];
            unsigned diff = 0u;
#pragma unroll
            for (int it = 0; it < TILE / 256; ++it) {
                const unsigned a0 = (unsigned)v[it].x >> 1, a1 = (unsigned)v[it].y >> 1, a2 = (unsigned)v[it].z >> 1, a3 = (unsigned)v[it].w >> 1;
                diff |= (pv[it].x ^ (a0 | (a1 << 16))) | (pv[it].y ^ (a2 | (a3 << 16)));
            }
            if (__any(diff != 0u)) {
#pragma unroll
                for (int it = 0; it < TILE / 256; ++it) {
                    const int64_t i0 = base + (int64_t)(it * 64 + lane) * 4;
                    track(i0, v[it].x); track(i0 + 1, v[it].y); track(i0 + 2, v[it].z); track(i0 + 3, v[it].w);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < TILE / 256; ++it) {
            const int b0 = __builtin_amdgcn_readfirstlane(v[it].x);
            const bool same = v[it].x == b0 && v[it].y == b0 && v[it].z == b0 && v[it].w == b0;
            if (__all(same)) {
                if (lane == 0 && (unsigned)b0 < (unsigned)nbins) cnt[b0] += 256;
            } else {
                // the usual case in point order is ONE cluster with its two sub-labels mixed: 256 atomics on two LDS addresses
                // serialise (the kernel spent 3/4 of its time there).  Count the first few distinct values of the wave with ballots
                // (a value costs four ballots and one LDS add), whatever is left goes through the atomics.
                const int vals[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
                unsigned todo = 0xFu;                                   // per lane: components not yet counted
                for (int round = 0; round < 4; ++round) {
                    const unsigned long long any = __ballot(todo != 0u);
                    if (!any) break;
                    const int leader = __ffsll((long long)any) - 1;
                    const unsigned tl = (unsigned)__shfl((int)todo, leader);
                    const int comp = __ffs((int)tl) - 1;
                    const int lv = comp == 0 ? vals[0] : comp == 1 ? vals[1] : comp == 2 ? vals[2] : vals[3];
                    const int bv = __shfl(lv, leader);
                    int c = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool hit = ((todo >> e) & 1u) && vals[e] == bv;
                        c += __popcll(__ballot(hit));
                        if (hit) todo &= ~(1u << e);
                    }
                    if (lane == 0 && (unsigned)bv < (unsigned)nbins) cnt[bv] += c;
                    if (c < 32) break;                                  // many different values (unsorted labels): the atomics are cheaper
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((todo >> e) & 1u) && (unsigned)vals[e] < (unsigned)nbins) atomicAdd(&cnt[vals[e]], 1);
            }
        }
    } else {
        for (int it = 0; it < TILE / 64; ++it) {
            const int64_t i = base + it * 64 + lane;
            if (i < n) {
                const int b = bins[i];
                if ((unsigned)b < (unsigned)nbins) atomicAdd(&cnt[b], 1);
                if (prev_lab) track(i, b);
            }
        }
    }
    }
    __syncthreads();
    if (tile < nt)
        for (int b = lane; b < nbins; b += 64) tile_hist[(int64_t)b * nt + tile] = cnt[b];
    if constexpr (SPEC) {
        if (tile_spec && tile < nt) {                  // (wave-uniform; everything below is wave-local: this wave's counters, no workgroup barrier)
            int *const spc = cnt_all + (W + wave) * nbins;
            const bool full = base + TILE <= n;
            auto onesided = [&](int bv) -> bool {
                if ((unsigned)bv >= (unsigned)nbins) return false;
                const int z2 = bv & ~1;
                return (cnt[z2] == 0) != (cnt[z2 + 1] == 0);
            };
            // does the tile hold a one-sided cluster at all?  One pass over the tile's counters (lane = bin); almost never: then nothing per point
            bool mine = false;
            for (int b = lane; b < nbins; b += 64) mine = mine || ((cnt[b] == 0) != (cnt[b ^ 1] == 0));
            const bool some = __any(mine);
            if (some) {
                for (int b = lane; b < nbins; b += 64) spc[b] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                auto redraw = [&](int64_t i, int bv) {
                    if (!onesided(bv)) return;
                    const Philox4 r = philox4x32_10(seed, (uint64_t)(first + i), epoch, STREAM_RESET);      // reset_recount_kernel's draw
                    const int nb = (bv & ~1) + (int)(r.v[0] & 1u);
                    spec_bins[i] = nb;                      // (what the scatter stores and places by if the cluster is flagged: no second draw there)
                    atomicAdd(&spc[nb], 1);
                };
                if (full) {
#pragma unroll
                    for (int it = 0; it < TILE / 256; ++it) {
                        const int64_t i0 = base + (int64_t)(it * 64 + lane) * 4;
                        redraw(i0, v[it].x); redraw(i0 + 1, v[it].y); redraw(i0 + 2, v[it].z); redraw(i0 + 3, v[it].w);
                    }
                } else {
                    for (int it = 0; it < TILE / 64; ++it) { const int64_t i = base + it * 64 + lane; if (i < n) redraw(i, bins[i]); }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
            if (some)          // (only the one-sided clusters' entries are ever read: scan_tiles_step_kernel)
                for (int b = lane; b < nbins; b += 64)
                    if ((cnt[b] == 0) != (cnt[b ^ 1] == 0)) tile_spec[(int64_t)b * nt + tile] = spc[b];
        }
    }
    if (fast_total)
        for (int b = threadIdx.x; b < nbins; b += 64 * W) {
            int v = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) v += cnt_all[w * nbins + b];
            if (v) atomicAdd(&fast_total[b * FAST_TOTAL_STRIDE], v);               // (a line per bin: 5 k tiles x ~3 bins on two lines took 18 us longer)
        }
}

// reset_bad_clusters! (src/local_clusters_actions.jl:501-516) + the re-count it makes necessary, one wave per sort tile: every workgroup
// derives the bad-cluster flags from the 2K sub-cluster occupancies (`totals`: this shard's, or `global_counts`: summed over the ranks),
// workgroup 0 publishes them (flags[0..K), flags[K] = any); a tile that holds points of a flagged cluster re-draws their sub-labels
// (Philox keyed by the global point index, as reset_sub_flagged_kernel) and re-counts itself -- tile_cnt of every other tile stays what
// the histogram wrote.  Replaces reset + second histogram + second scan of the per-step pass (three launches fewer).
// `cside` (nullable; the one-collective pass of a multi-rank run, run_stats in dpmm_api.cpp): the GLOBAL occupancies are not known yet, so
// the reset is applied SPECULATIVELY to this shard's candidates -- clusters with exactly one empty sub-cluster HERE (a cluster that is bad
// globally is one of them on every rank that holds points of it) -- and cside[k] records which side the shard's points were on (1: all
// left, 2: all right, 0: not a candidate): enough to rebuild the rows of the labels as swept (the non-empty side carries left' + right')
// and to undo the reset of a candidate that turns out not to be bad (all of its points go back to that side).  flags[] then holds the
// candidates, not the verdict: niw_finalize_rows_kernel writes the verdict behind the all-reduce.
template <int TILE>
__global__ __launch_bounds__(TILE / 8) void reset_recount_kernel(int32_t *__restrict__ bins, int64_t n, int64_t first, int nbins, int nt,
                                                                 const int32_t *__restrict__ totals, const long long *__restrict__ global_counts,
                                                                 int32_t *__restrict__ tile_cnt, uint8_t *__restrict__ flags, int K, uint64_t seed, uint32_t epoch,
                                                                 uint8_t *__restrict__ cside) {
    // TILE / 8 threads per tile, eight points each (round 6: one wave per 2048-point tile drew 32 Philox blocks per lane one after the other -- the
    // launch lasted as long as ONE tile of a flagged cluster, 17-21 us at N = 1e7; four waves share it now)
    constexpr int NT = TILE / 8, PER = 8;
    extern __shared__ int cnt[];                 // [nbins] counters | [K] flag bytes
    uint8_t *f = reinterpret_cast<uint8_t *>(cnt + nbins);
    const int tid = threadIdx.x;
    bool anyl = false;
    for (int k = tid; k < K; k += NT) {
        const long long a = global_counts ? global_counts[2 * k] : (long long)totals[(2 * k) * FAST_TOTAL_STRIDE];
        const long long b = global_counts ? global_counts[2 * k + 1] : (long long)totals[(2 * k + 1) * FAST_TOTAL_STRIDE];
        const bool bad = cside ? ((a == 0) != (b == 0)) : (a == 0 || b == 0);
        f[k] = bad ? 1 : 0;
        anyl = anyl || bad;
        if (blockIdx.x == 0) {
            flags[k] = bad ? 1 : 0;
            if (cside) cside[k] = bad ? (b == 0 ? 1 : 2) : 0;
        }
    }
    const bool any = __syncthreads_or(anyl ? 1 : 0) != 0;        // (also publishes f[] to the workgroup)
    if (blockIdx.x == 0 && tid == 0) flags[K] = any ? 1 : 0;
    if (!any) return;
    // does this tile hold a point of a flagged cluster at all?  The histogram's counts say so without reading the tile's labels (round 6: with the
    // points of a component contiguous in storage -- the reference generator's layout -- 94 % of the tiles leave here)
    {
        bool need = false;
        for (int k = tid; k < K; k += NT)
            if (f[k]) need = need || (tile_cnt[(int64_t)(2 * k) * nt + blockIdx.x] | tile_cnt[(int64_t)(2 * k + 1) * nt + blockIdx.x]) != 0;
        if (!__syncthreads_or(need ? 1 : 0)) return;
    }
    const int64_t base = (int64_t)blockIdx.x * TILE;
    int v[PER];
    bool hit = false;
    const bool full = base + TILE <= n;       // full tile: 16-byte loads (a thread owns four consecutive points per trip), else one point per trip
    if (full) {
        const int4 *src = reinterpret_cast<const int4 *>(bins + base);
#pragma unroll
        for (int it = 0; it < PER / 4; ++it) {
            const int4 q = src[it * NT + tid];
            v[4 * it] = q.x; v[4 * it + 1] = q.y; v[4 * it + 2] = q.z; v[4 * it + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int64_t i = base + it * NT + tid;
            v[it] = i < n ? bins[i] : -1;
        }
    }
#pragma unroll
    for (int it = 0; it < PER; ++it) {
        const int z = v[it] >> 1;
        hit = hit || (v[it] >= 0 && z < K && f[z]);
    }
    if (!__syncthreads_or(hit ? 1 : 0)) return;
    for (int b = tid; b < nbins; b += NT) cnt[b] = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < PER; ++it) {
        const int64_t i = full ? base + (int64_t)((it >> 2) * NT + tid) * 4 + (it & 3) : base + it * NT + tid;
        int bv = v[it];
        const int z = bv >> 1;
        if (bv >= 0 && z < K && f[z]) {
            const Philox4 r = philox4x32_10(seed, (uint64_t)(first + i), epoch, STREAM_RESET);
            bv = 2 * z + (int)(r.v[0] & 1u);
            bins[i] = bv;
        }
        if ((unsigned)bv < (unsigned)nbins) atomicAdd(&cnt[bv], 1);
    }
    __syncthreads();
    for (int b = tid; b < nbins; b += NT) tile_cnt[(int64_t)b * nt + blockIdx.x] = cnt[b];
}

// exclusive scan over the tiles of one bin (in place) + bin total
template <class Src>
__device__ __forceinline__ void scan_one_bin_of(Src src, int32_t *__restrict__ tile_hist, int nt, int32_t *__restrict__ bin_total, int *part) {
    int32_t *row = tile_hist + (int64_t)blockIdx.x * nt;
    const int per = (nt + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(lo + per, nt);
    int s = 0;
    // a thread's counts stay in registers between the two passes when they fit (up to 24 per thread: 6144 tiles = 12.6e6 points per shard at
    // 2048-point tiles): the second pass then reads nothing -- the flagged clusters' source is three loads per element
    constexpr int KEEP = 24;
    int kept[KEEP];
    const bool keep = per <= KEEP;
    if (keep) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u) { kept[u] = (u < per && lo + u < hi) ? src(lo + u) : 0; s += kept[u]; }
    } else
    for (int i = lo; i < hi; ++i) s += src(i);
    // inclusive scan over the 256 partials: inside each wave with shuffles (no barrier), the four wave totals through LDS (ONE barrier;
    // the Hillis-Steele loop this replaces had sixteen)
    int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(inc, off);
        if ((threadIdx.x & 63) >= off) inc += v;
    }
    if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = inc;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const int before = (w > 0 ? part[0] : 0) + (w > 1 ? part[1] : 0) + (w > 2 ? part[2] : 0);
    inc += before;
    if (threadIdx.x == 255) part[255] = inc;       // (the total, where the callers read it)
    int run = inc - s;  // exclusive prefix of this thread's range
    if (keep) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u)
            if (u < per && lo + u < hi) { row[lo + u] = run; run += kept[u]; }
    } else
    for (int i = lo; i < hi; ++i) {
        const int v = src(i);
        row[i] = run;
        run += v;
    }
    if (threadIdx.x == 255) bin_total[blockIdx.x] = inc;
}
__device__ __forceinline__ void scan_one_bin(const int32_t *__restrict__ tile_cnt, int32_t *__restrict__ tile_hist, int nt,
                                             int32_t *__restrict__ bin_total, int *part) {
    const int32_t *src = tile_cnt + (int64_t)blockIdx.x * nt;
    scan_one_bin_of([src](int i) -> int { return src[i]; }, tile_hist, nt, bin_total, part);
}
__global__ __launch_bounds__(256) void scan_tiles_kernel(const int32_t *__restrict__ tile_cnt, int32_t *__restrict__ tile_hist, int nt,
                                                         int32_t *__restrict__ bin_total) {
    __shared__ int part[256];
    scan_one_bin(tile_cnt, tile_hist, nt, bin_total, part);
}
// The scan of the per-step pass when the histogram counted the bad-cluster reset ahead (hist_kernel<.., SPEC>): the workgroup of bin b derives
// its cluster's flag from the 2K sub-cluster occupancies exactly as reset_recount_kernel does (`totals`: this shard's; `global_counts`: summed
// over the ranks; `cside` != null: the speculative form of the one-collective pass -- candidates, and which side their points were on) and
// scans tile_spec for a flagged cluster, tile_cnt otherwise.  Workgroup 0 publishes flags[0..K), flags[K] = any, cside[].
__global__ __launch_bounds__(256) void scan_tiles_step_kernel(const int32_t *__restrict__ tile_cnt, const int32_t *__restrict__ tile_spec, int32_t *__restrict__ tile_hist,
                                                              int nt, int32_t *__restrict__ bin_total, const int32_t *__restrict__ totals,
                                                              const long long *__restrict__ global_counts, uint8_t *__restrict__ flags, int K,
                                                              uint8_t *__restrict__ cside) {
    __shared__ int part[256];
    __shared__ int any_s;
    auto occupancy = [&](int k, long long &a, long long &b) {
        a = global_counts ? global_counts[2 * k] : (long long)totals[(2 * k) * FAST_TOTAL_STRIDE];
        b = global_counts ? global_counts[2 * k + 1] : (long long)totals[(2 * k + 1) * FAST_TOTAL_STRIDE];
    };
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) any_s = 0;
        __syncthreads();
        bool anyl = false;
        for (int k = threadIdx.x; k < K; k += 256) {
            long long a, b;
            occupancy(k, a, b);
            const bool bad = cside ? ((a == 0) != (b == 0)) : (a == 0 || b == 0);
            flags[k] = bad ? 1 : 0;
            if (cside) cside[k] = bad ? (b == 0 ? 1 : 2) : 0;
            anyl = anyl || bad;
        }
        if (anyl) any_s = 1;              // (benign race: every writer stores 1)
        __syncthreads();
        if (threadIdx.x == 0) flags[K] = any_s ? 1 : 0;
    }
    long long a, b;
    occupancy((int)blockIdx.x >> 1, a, b);
    const bool bad = cside ? ((a == 0) != (b == 0)) : (a == 0 || b == 0);
    if (!bad) { scan_one_bin(tile_cnt, tile_hist, nt, bin_total, part); return; }
    // a flagged cluster: one of its sub-bins is empty in every tile, so every tile that holds points of it counted the re-draw ahead (tile_spec
    // is written for exactly those (tile, cluster) pairs); a tile without points of it wrote nothing and contributes nothing
    const int32_t *c0 = tile_cnt + (int64_t)blockIdx.x * nt, *c1 = tile_cnt + (int64_t)(blockIdx.x ^ 1) * nt, *sp = tile_spec + (int64_t)blockIdx.x * nt;
    scan_one_bin_of([c0, c1, sp](int i) -> int { return (c0[i] | c1[i]) != 0 ? sp[i] : 0; }, tile_hist, nt, bin_total, part);
}

// bin_start[b] = sum_{b'<b} total ; item_start[b] = sum_{b'<b} ceil(sel*total / chunk)
__device__ __forceinline__ void starts_body(const int32_t *bin_total, const uint8_t *bin_sel,
                                            int nbins, int chunk, int32_t *__restrict__ bin_start,
                                            int32_t *__restrict__ item_start, int32_t *__restrict__ perm_total, int *pa, int *pb) {
    const int per = (nbins + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(lo + per, nbins);
    int sa = 0, sb = 0;
    for (int b = lo; b < hi; ++b) {
        const int t = bin_total[b];
        sa += t;
        sb += bin_sel[b] ? (t + chunk - 1) / chunk : 0;
    }
    int ia = sa, ib = sb;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int va = __shfl_up(ia, off), vb = __shfl_up(ib, off);
        if ((threadIdx.x & 63) >= off) { ia += va; ib += vb; }
    }
    __syncthreads();                                   // (pa / pb may still be read by the caller's previous use)
    if ((threadIdx.x & 63) == 63) { pa[threadIdx.x >> 6] = ia; pb[threadIdx.x >> 6] = ib; }
    __syncthreads();
    {
        const int w = threadIdx.x >> 6;
        ia += (w > 0 ? pa[0] : 0) + (w > 1 ? pa[1] : 0) + (w > 2 ? pa[2] : 0);
        ib += (w > 0 ? pb[0] : 0) + (w > 1 ? pb[1] : 0) + (w > 2 ? pb[2] : 0);
    }
    __syncthreads();
    if (threadIdx.x == 255) { pa[255] = ia; pb[255] = ib; }
    __syncthreads();
    int ra = ia - sa, rb = ib - sb;
    for (int b = lo; b < hi; ++b) {
        bin_start[b] = ra;
        item_start[b] = rb;
        const int t = bin_total[b];
        ra += t;
        rb += bin_sel[b] ? (t + chunk - 1) / chunk : 0;
    }
    if (threadIdx.x == 255) {
        bin_start[nbins] = pa[255];
        item_start[nbins] = pb[255];
        if (perm_total) *perm_total = pa[255];
    }
}
__global__ __launch_bounds__(256) void starts_kernel(const int32_t *__restrict__ bin_total, const uint8_t *__restrict__ bin_sel,
                                                     int nbins, int chunk, int32_t *__restrict__ bin_start,
                                                     int32_t *__restrict__ item_start, int32_t *__restrict__ perm_total) {
    __shared__ int pa[256], pb[256];
    starts_body(bin_total, bin_sel, nbins, chunk, bin_start, item_start, perm_total, pa, pb);
}
// The starts of the per-step pass, one workgroup behind scan_tiles_kernel: bin_start / item_start from the bin totals; it also clears the
// histogram's running totals for the next pass and decides, per cluster, which bins the statistics kernels compute
// (bin_sel) -- mode[k] = 0: both sub-clusters (the cluster was touched since its cached row was formed, or is empty, or force_all),
// 1: only the right one (the left is the larger: derived as cache - right), 2: only the left one.
// (Rounds 2 - 4 had scan + starts in ONE launch, the workgroup that finished last doing this part: the ticket needs an agent-scope release fence
// in every workgroup, which on this part is an L2 write-back each -- 13 - 17 us for the one launch against 5 + 5 for the two.)
__global__ __launch_bounds__(256) void starts_step_kernel(int32_t *bin_total, uint8_t *bin_sel, int nbins, int chunk,
                                                          int32_t *__restrict__ bin_start, int32_t *__restrict__ item_start,
                                                          int32_t *__restrict__ perm_total, int32_t *__restrict__ fast_total,
                                                          uint8_t *__restrict__ mode, const uint8_t *__restrict__ dirty, int force_all) {
    __shared__ int part[256], pb[256];
    if (fast_total) for (int b = threadIdx.x; b < nbins; b += 256) fast_total[b * FAST_TOTAL_STRIDE] = 0;
    if (mode) {
        const bool all = force_all || dirty[DPMM_MAX_CLUSTERS_K];
        for (int k = threadIdx.x; k < nbins / 2; k += 256) {
            const int nl = bin_total[2 * k], nr = bin_total[2 * k + 1];
            const int m = (all || dirty[k] || nl + nr == 0) ? 0 : (nl <= nr ? 2 : 1);
            mode[k] = (uint8_t)m;
            bin_sel[2 * k] = m != 1;
            bin_sel[2 * k + 1] = m != 2;
        }
    }
    __syncthreads();
    starts_body(bin_total, bin_sel, nbins, chunk, bin_start, item_start, perm_total, part, pb);
}

// STEP (the per-step pass, round 6): the starts of starts_step_kernel are computed HERE -- every tile's wave forms the exclusive prefix of the
// 2K bin totals itself (one wave-level scan per 64 bins: K = 32 is one), workgroup 0 also publishes what the later launches read (bin_start,
// item_start, perm_total, the statistics' bin selection and modes) and clears the histogram's running totals for the next pass.  One launch
// less in the chain sweep -> statistics (4.6 us at its latency floor, whatever n).  Same values: integer prefix sums.
struct StepStarts {
    const int32_t *bin_total; uint8_t *bin_sel; int chunk; int32_t *bin_start_out; int32_t *item_start; int32_t *perm_total; int32_t *fast_total;
    uint8_t *mode; const uint8_t *dirty; int force_all;
    const uint8_t *reset_flags; const int32_t *spec_bins;      // (reset_flags != null: apply the bad-cluster reset while placing -- spec_bins[i]: the histogram's re-draw of point i)
    const int32_t *tile_cnt;                                    // (the histogram's counts: does this tile hold a point of a flagged cluster at all?)
};
template <int TILE, bool STEP>
__global__ __launch_bounds__(64) void scatter_kernel(int32_t *bins, int64_t n, int nbins, int nt,
                                                     const int32_t *__restrict__ tile_hist,
                                                     const int32_t *__restrict__ bin_start, int32_t *__restrict__ perm, StepStarts st) {
    extern __shared__ int base[];
    const int lane = threadIdx.x;
    int nbits = 0;
    while ((1 << nbits) < nbins) ++nbits;
    // STEP with st.reset_flags: the bad-cluster reset is APPLIED here (the histogram counted it ahead, the scan chose those counts): a point of a
    // flagged cluster gets the sub-label of reset_recount_kernel's draw, is stored, and is placed by it
    uint8_t *const rflag = reinterpret_cast<uint8_t *>(base + nbins);
    bool do_reset = false;
    if constexpr (STEP) {
        if (st.reset_flags && st.reset_flags[nbins >> 1]) {           // flags[K]: any cluster flagged (workgroup-uniform)
            // ... and only a tile that holds points of a flagged cluster looks at its points' flags (in storage order most tiles hold none)
            bool need = false;
            for (int b = lane; b < nbins; b += 64) {
                const uint8_t f = st.reset_flags[b >> 1];
                if (!(b & 1)) rflag[b >> 1] = f;
                need = need || (f && st.tile_cnt[(int64_t)b * nt + blockIdx.x] > 0);
            }
            do_reset = __any(need);
        }
    }
    if constexpr (STEP) {
        const bool pub = blockIdx.x == 0;
        const bool all = pub && st.mode && (st.force_all || st.dirty[DPMM_MAX_CLUSTERS_K]);
        int carry = 0, icarry = 0;
        for (int b0 = 0; b0 < nbins; b0 += 64) {
            const int b = b0 + lane;
            const int t = b < nbins ? st.bin_total[b] : 0;
            int inc = t;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
            const int excl = carry + inc - t;
            if (b < nbins) base[b] = excl + tile_hist[(int64_t)b * nt + blockIdx.x];
            carry += __shfl(inc, 63);
            if (pub) {
                int sel = 0;
                if (b < nbins) {
                    if (st.mode) {
                        const int k = b >> 1;
                        const int nl = st.bin_total[2 * k], nr = st.bin_total[2 * k + 1];
                        const int m = (all || st.dirty[k] || nl + nr == 0) ? 0 : (nl <= nr ? 2 : 1);
                        sel = (b & 1) ? (m != 2) : (m != 1);
                        if (!(b & 1)) st.mode[k] = (uint8_t)m;
                        st.bin_sel[b] = (uint8_t)sel;
                    } else sel = st.bin_sel[b];
                }
                const int items = sel ? (t + st.chunk - 1) / st.chunk : 0;
                int iinc = items;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(iinc, off); if (lane >= off) iinc += v; }
                if (b < nbins) {
                    st.bin_start_out[b] = excl;
                    st.item_start[b] = icarry + iinc - items;
                    if (st.fast_total) st.fast_total[b * FAST_TOTAL_STRIDE] = 0;
                }
                icarry += __shfl(iinc, 63);
            }
        }
        if (pub && lane == 0) {
            st.bin_start_out[nbins] = carry;
            st.item_start[nbins] = icarry;
            if (st.perm_total) *st.perm_total = carry;
        }
    } else {
        for (int b = lane; b < nbins; b += 64) base[b] = bin_start[b] + tile_hist[(int64_t)b * nt + blockIdx.x];
    }
    __syncthreads();
    const int64_t tbase = (int64_t)blockIdx.x * TILE;
    // all loads of the tile in flight before the first is used (one wave per tile: a load per trip would serialise 32 latencies)
    int bv[TILE / 64];
#pragma unroll
    for (int it = 0; it < TILE / 64; ++it) {
        const int64_t i = tbase + it * 64 + lane;
        bv[it] = bins[i < n ? i : n - 1];
    }
#pragma unroll
    for (int it = 0; it < TILE / 64; ++it) {
        const int64_t i = tbase + it * 64 + lane;
        int b = i < n ? bv[it] : -1;
        if ((unsigned)b >= (unsigned)nbins) b = -1;
        if constexpr (STEP) {
            if (do_reset && b >= 0 && rflag[b >> 1]) {
                b = st.spec_bins[i];             // (the histogram's re-draw of this point: its cluster was one-sided in the point's tile -- every flagged cluster is, in every tile)
                bins[i] = b;
            }
        }
        const bool valid = b >= 0;
        // lanes with the same bin, by one ballot per bit of the bin id (a fixed ceil(log2 nbins) steps; the leader-by-leader loop it
        // replaces took one step per DISTINCT bin in the wave: ~40 on unsorted labels, e.g. bag-of-words data); a wave with one
        // common bin (neighbours share a label after an ordered sweep) needs none
        unsigned long long m = __ballot(valid);
        const int bfirst = __builtin_amdgcn_readfirstlane(b);
        if (!__all(b == bfirst)) {
            // two distinct values (one cluster, its two sub-labels mixed -- the usual case in point order): two ballots
            const unsigned long long m0 = __ballot(b == bfirst), rest = m & ~m0;
            const int bsecond = __shfl(b, rest ? __ffsll((long long)rest) - 1 : 0);
            const unsigned long long m1 = __ballot(valid && b == bsecond);
            if (((m0 & m) | m1) == m) {
                m = (b == bfirst) ? (m0 & m) : m1;
            } else {
                for (int bit = 0; bit < nbits; ++bit) {
                    const unsigned long long bal = __ballot(valid && ((b >> bit) & 1));
                    m &= ((b >> bit) & 1) ? bal : ~bal;
                }
            }
        }
        const int rank = __popcll(m & ((1ull << lane) - 1ull)), cntb = __popcll(m);
        int pos = 0;
        if (valid) pos = base[b] + rank;
        // (one wave per workgroup: LDS operations of a wave complete in order -- a wave-level fence instead of a workgroup barrier)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (valid) {
            perm[pos] = (int32_t)i;
            if (rank == 0) base[b] += cntb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}

// After the reduce of a per-step pass: the rows of the sub-clusters that were not computed, and the cache they come from.
//   mode[k] = 0: both rows were computed -> cache[k] = left + right (the cluster-level statistics, valid until a point enters or leaves k)
//   mode[k] = 1 / 2: left / right = cache[k] - the computed row.  The derived side is the LARGER one: its relative error is that of the
//   cached sum (~1e-16); N is an integer count and exact either way.  Clears the dirty flags for the next pass.
__global__ __launch_bounds__(256) void derive_rows_kernel(double *__restrict__ out, double *__restrict__ cache, const uint8_t *__restrict__ mode,
                                                          uint8_t *__restrict__ dirty, int64_t stride, int K,
                                                          const uint8_t *__restrict__ flags_src, uint8_t *__restrict__ flags_dst) {
    // rider: the bad-cluster flags of the pass go to the host's pinned block from here (one launch less than a copy kernel of their own)
    if (flags_dst && blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i <= K; i += 256) flags_dst[i] = flags_src[i];
    const int k = blockIdx.y;
    const int64_t e = blockIdx.x * 256ll + threadIdx.x;
    const int m = mode[k];
    if (e < stride) {
        double *l = out + (int64_t)(2 * k) * stride, *r = l + stride, *c = cache + (int64_t)k * stride;
        if (m == 0) c[e] = l[e] + r[e];
        else if (m == 1) l[e] = c[e] - r[e];
        else r[e] = c[e] - l[e];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { dirty[k] = 0; if (k == 0) dirty[DPMM_MAX_CLUSTERS_K] = 0; }
}
hipError_t launch_derive_rows(double *out, double *cache, const uint8_t *mode, uint8_t *dirty, int64_t stride, int K, const uint8_t *flags_src,
                              uint8_t *flags_dst, hipStream_t s) {
    DPMM_LAUNCH(derive_rows_kernel, dim3((unsigned)((stride + 255) / 256), K), dim3(256), 0, s, out, cache, mode, dirty, stride, K, flags_src, flags_dst);
    return hipGetLastError();
}

// Sort tile = points per sorting wave (SortBufs::tile): 2048 for big shards, 512 below ~4e6 points -- at the 8-GPU shard size the three
// tile kernels are latency chains of one wave per tile (8 / 32 dependent trips per wave at 2048 points: 12 us each for 5 MB of labels).
#define DPMM_TILE_DISPATCH(tile, CALL512, CALL2048) do { if ((tile) == 512) { CALL512; } else { CALL2048; } } while (0)
static inline int sort_nt(int64_t n, const SortBufs &b) { return (int)((n + b.tile - 1) / b.tile); }
template <int TILE>
static inline void launch_hist(const int32_t *bins, int64_t n, int nbins, int nt, int32_t *tile_cnt, int32_t *fast_total, uint16_t *prev_lab, uint8_t *dirty,
                               hipStream_t s, int32_t *tile_spec = nullptr, int32_t *spec_bins = nullptr, int64_t first = 0, uint64_t seed = 0, uint32_t epoch = 0) {
    // eight tiles per workgroup while eight count arrays fit comfortably in LDS (K <= 512), else one
    if (tile_spec && nbins <= STEP_SPEC_MAX_BINS)       // (a second set of eight counter arrays: 32 KiB at 512 bins)
        DPMM_LAUNCH((hist_kernel<TILE, 8, true>), dim3((nt + 7) / 8), dim3(512), 16 * nbins * sizeof(int), s, bins, n, nbins, nt, tile_cnt, fast_total, prev_lab, dirty,
                    tile_spec, spec_bins, first, seed, epoch);
    else if (nbins <= 1024) DPMM_LAUNCH((hist_kernel<TILE, 8, false>), dim3((nt + 7) / 8), dim3(512), 8 * nbins * sizeof(int), s, bins, n, nbins, nt, tile_cnt, fast_total, prev_lab, dirty,
                                        (int32_t *)nullptr, (int32_t *)nullptr, (int64_t)0, (uint64_t)0, 0u);
    else DPMM_LAUNCH((hist_kernel<TILE, 1, false>), dim3(nt), dim3(64), nbins * sizeof(int), s, bins, n, nbins, nt, tile_cnt, fast_total, prev_lab, dirty,
                     (int32_t *)nullptr, (int32_t *)nullptr, (int64_t)0, (uint64_t)0, 0u);
}
hipError_t launch_sort_by_bin(const int32_t *bins, int64_t n, int nbins, const SortBufs &b, hipStream_t s) {
    const int nt = sort_nt(n, b);
    if (nt == 0) return hipSuccess;
    DPMM_TILE_DISPATCH(b.tile,
        launch_hist<512>(bins, n, nbins, nt, b.tile_cnt, (int32_t *)nullptr, (uint16_t *)nullptr, (uint8_t *)nullptr, s),
        launch_hist<2048>(bins, n, nbins, nt, b.tile_cnt, (int32_t *)nullptr, (uint16_t *)nullptr, (uint8_t *)nullptr, s));
    DPMM_LAUNCH(scan_tiles_kernel, dim3(nbins), dim3(256), 0, s, b.tile_cnt, b.tile_hist, nt, b.bin_total);
    return hipGetLastError();
}
// The sort of the per-step pass in four launches (n > 0): histogram (+ running totals) -> [caller: all-reduce of the totals] ->
// launch_step_reset (flags, sub-label reset, re-count of the touched tiles) -> launch_step_scan_scatter (scan + starts, scatter).
// spec (DPMM_OPT_CHAIN_FUSION bit 8, nbins <= STEP_SPEC_MAX_BINS): the tiles also count the outcome of the bad-cluster reset ahead (hist_kernel<.., SPEC>)
hipError_t launch_step_hist(const int32_t *bins, int64_t n, int nbins, const SortBufs &b, hipStream_t s, int spec, int64_t first, uint64_t seed, uint32_t epoch) {
    const int nt = sort_nt(n, b);
    uint8_t *dirty = b.prev_lab ? b.cdirty : (uint8_t *)nullptr;
    int32_t *ts = (spec && b.tile_spec && b.spec_bins && nbins <= STEP_SPEC_MAX_BINS) ? b.tile_spec : (int32_t *)nullptr;
    DPMM_TILE_DISPATCH(b.tile,
        launch_hist<512>(bins, n, nbins, nt, b.tile_cnt, b.fast_total, b.prev_lab, dirty, s, ts, b.spec_bins, first, seed, epoch),
        launch_hist<2048>(bins, n, nbins, nt, b.tile_cnt, b.fast_total, b.prev_lab, dirty, s, ts, b.spec_bins, first, seed, epoch));
    return hipGetLastError();
}
hipError_t launch_step_reset(int32_t *bins, int64_t n, int64_t first, int nbins, const SortBufs &b, const long long *global_counts, uint8_t *flags,
                             int K, uint64_t seed, uint32_t epoch, uint8_t *cside, hipStream_t s) {
    const int nt = sort_nt(n, b);
    const size_t lds = nbins * sizeof(int) + ((K + 3) & ~3);
    DPMM_TILE_DISPATCH(b.tile,
        DPMM_LAUNCH(reset_recount_kernel<512>, dim3(nt), dim3(64), lds, s, bins, n, first, nbins, nt, b.fast_total, global_counts, b.tile_cnt, flags, K, seed, epoch, cside),
        DPMM_LAUNCH(reset_recount_kernel<2048>, dim3(nt), dim3(256), lds, s, bins, n, first, nbins, nt, b.fast_total, global_counts, b.tile_cnt, flags, K, seed, epoch, cside));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ items
__device__ __forceinline__ int find_bin(const int32_t *__restrict__ item_start, int nbins, int item) {
    int lo = 0, hi = nbins;  // largest b with item_start[b] <= item and item < item_start[b+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (item_start[mid] <= item) lo = mid; else hi = mid;
    }
    // skip empty bins that share the same start
    while (lo + 1 < nbins && item_start[lo + 1] <= item) ++lo;
    return lo;
}

// ------------------------------------------------------------------------------------ NIW statistics
#ifndef DPMM_STATS16_SB
#define DPMM_STATS16_SB 4       // k-steps per LDS batch of the D > 128 kernel (two buffers: 64 KiB of LDS; 2: 1.5 % slower)
#endif
template <int NBK>
struct StatCfg {
    static constexpr int NPAIR = NBK * (NBK + 1) / 2;
    static constexpr int DP = 16 * NBK;
    // NBK = 16: EIGHT panels of 17 pairs -- 136 accumulator registers per wave, two waves per SIMD.  (Four panels of 34 need 272: more than
    // the 256-register accumulation file; with the shared data path of niw_stats_body16 that variant spilled 691 registers.  Eight panels
    // with every wave loading and converting for itself had measured the same as four: 1.13 ms.)
    static constexpr int NPANEL = (NBK <= 4) ? 1 : (NBK == 8 ? 4 : 8);      // NBK = 8: four panels of 9 pairs on the shared data path (two of 18, each
                                                                              // wave loading and converting for itself, before)
    static constexpr int PP = (NPAIR + NPANEL - 1) / NPANEL;  // pairs per panel
};

__host__ __device__ inline int64_t niw_slab_stride_nbk(int NBK) { return (int64_t)(NBK * (NBK + 1) / 2) * 256 + 16 * NBK; }

template <int NBK, int PANEL>
__device__ __forceinline__ void niw_stats_body(const StatsArgs &A, int seg, int cnt, double *__restrict__ slab) {
    using C = StatCfg<NBK>;
    constexpr int P0 = PANEL * C::PP;
    constexpr int P1 = (P0 + C::PP < C::NPAIR) ? P0 + C::PP : C::NPAIR;
    constexpr int NP = P1 - P0;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    f64x4 acc[NP > 0 ? NP : 1];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = (f64x4){0., 0., 0., 0.};
    double xs[NBK];
#pragma unroll
    for (int b = 0; b < NBK; ++b) xs[b] = 0.;

    constexpr int U = (NBK <= 4) ? 4 : (NBK <= 8 ? 2 : 1);  // k-steps (of 4 points) per batch
    const int nsteps = (cnt + 3) >> 2;
    const int nbatch = (nsteps + U - 1) / U;
    // Loads are UNCONDITIONAL (clamped addresses) and nothing touches a loaded value before its batch is consumed:
    // a conditional load becomes a branch with `s_waitcnt vmcnt(0)` behind it, and a select right after a load
    // waits for it -- either drains every prefetch in flight.  Rows beyond the item and columns beyond the row
    // are zeroed when the batch is converted to Float64.
    auto load_idx = [&](int b, int (&pt)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) pt[u] = A.sb.perm[seg + min(4 * (b * U + u) + g, cnt - 1)];
    };
    bool colok[NBK >= 4 ? NBK / 4 : NBK];
    int coloff[NBK >= 4 ? NBK / 4 : NBK];
#pragma unroll
    for (int c = 0; c < (NBK >= 4 ? NBK / 4 : NBK); ++c) {
        const int col = NBK * i + (NBK >= 4 ? 4 * c : c);
        colok[c] = col < A.ldx;
        coloff[c] = colok[c] ? col : 0;
    }
    const bool allcols = A.ldx >= 16 * NBK;        // no column of the padded width lies beyond the row
    auto load_x = [&](const int (&pt)[U], float (&xf)[U][NBK]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int pv = pt[u];
            asm volatile("" : "+v"(pv));     // opaque here: the widening of a freshly loaded index must not be hoisted to its load
            const float *xrow = A.X + (int64_t)pv * A.ldx;
            if constexpr (NBK >= 4) {
#pragma unroll
                for (int c4 = 0; c4 < NBK / 4; ++c4) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(xrow + coloff[c4]);
                    xf[u][4 * c4 + 0] = v.x; xf[u][4 * c4 + 1] = v.y; xf[u][4 * c4 + 2] = v.z; xf[u][4 * c4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int b2 = 0; b2 < NBK; ++b2) xf[u][b2] = xrow[coloff[b2]];
            }
        }
    };
    // Software pipeline: x two batches ahead of the MFMAs, perm indices one batch further.  (D = 256 used to run ONE one-k-step
    // batch ahead: 1 088 matrix-pipe cycles of cover against > 2 000 cycles of HBM latency at one wave per SIMD; a third batch
    // ahead spills -- every buffer role costs ~40 VGPRs, addresses included, and 256 of the 512 registers are accumulators.)
    // In every iteration the index loads are issued BEFORE the x loads that consume the previous iteration's indices: vmcnt
    // retires in order, so waiting for an index never drains the x loads behind it.
    constexpr int AHEAD = 2;
    constexpr int NBUF = AHEAD + 1;
    // The x buffers rotate by ROLE (loop unrolled over the buffers), never by copying: a register move of a
    // freshly loaded value would wait for the load it was meant to overlap.
    int pt_b[U], pt_c[U];
    float xbuf[NBUF][U][NBK];
    auto step = [&](int bt, const float (&xu)[U][NBK], float (&xl)[U][NBK]) {
        // scheduling fences keep the issue order idx -> x -> MFMAs: left alone, the scheduler hoists the next step's address
        // arithmetic (which needs the newest indices) into this step and the wait for them drains the x loads as well
        load_idx(bt + AHEAD + 1, pt_c);
        __builtin_amdgcn_sched_barrier(0);
        load_x(pt_b, xl);                          // batch bt + AHEAD
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool rowok = 4 * (bt * U + u) + g < cnt;
            double xd[NBK];
            // whole k-step inside the item and no padded columns (wave-uniform, the common case): plain conversions -- the masks cost two
            // instructions per element and this kernel issues ~3.7 vector instructions per matrix instruction as it is
            if (allcols && 4 * (bt * U + u) + 3 < cnt) {
#pragma unroll
                for (int b2 = 0; b2 < NBK; ++b2) xd[b2] = (double)xu[u][b2];
            } else {
#pragma unroll
                for (int b2 = 0; b2 < NBK; ++b2) {
                    const bool keep = rowok && colok[NBK >= 4 ? b2 / 4 : b2];
                    xd[b2] = (double)(keep ? xu[u][b2] : 0.f);
                }
            }
            if constexpr (PANEL == 0) {
#pragma unroll
                for (int b2 = 0; b2 < NBK; ++b2) xs[b2] += xd[b2];
            }
            // lower block triangle, pair index p(ba,bb) = ba(ba+1)/2 + bb, ba >= bb
#pragma unroll
            for (int ba = 0; ba < NBK; ++ba)
#pragma unroll
                for (int bb = 0; bb <= ba; ++bb) {
                    const int p = ba * (ba + 1) / 2 + bb;
                    if (p >= P0 && p < P1)
                        acc[p - P0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[ba], xd[bb], acc[p - P0], 0, 0, 0);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) pt_b[u] = pt_c[u];   // indices were issued before this step's x loads: no drain
        __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: batches 0 .. AHEAD-1 in flight, indices of batch AHEAD loaded
    load_idx(0, pt_b);
#pragma unroll
    for (int r = 0; r < AHEAD; ++r) {
        load_x(pt_b, xbuf[r]);
        load_idx(r + 1, pt_b);
    }
    for (int bt = 0; bt < nbatch; bt += NBUF) {          // steps beyond nbatch see masked rows: zero contribution, no branch
#pragma unroll
        for (int r = 0; r < NBUF; ++r) step(bt + r, xbuf[r], xbuf[(r + AHEAD) % NBUF]);
    }
    // slab[pair][r][lane]
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(int64_t)(P0 + p) * 256 + r * 64 + lane] = acc[p][r];
    if constexpr (PANEL == 0) {
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            double v = xs[b];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) slab[(int64_t)C::NPAIR * 256 + NBK * i + b] = v;
        }
    }
}

// D > 64 (NBK = 16: eight panels = eight waves, two per SIMD; NBK = 8: four panels): the panels SHARE the data path.  The threads load a batch of SB k-steps (4 SB points)
// once, convert to Float64 once (rows beyond the item and columns beyond the row zeroed there) and store it to LDS in operand order
// [block b][point p][lane column i]; a wave then fetches the 16 operands of a k-step with 16 conflict-free ds_read_b64 and issues its
// 17 matrix instructions.  Before, every wave loaded and converted everything itself: 126 vector instructions per 34 matrix
// instructions, ~110 of them accumulator-file moves (272 accumulator registers + three x buffers + the converted row do not fit next to
// each other), matrix pipe busy 55 %.  One barrier per batch; the global loads of batch n + 1 are issued before the matrix phase of
// batch n and consumed after it.  Column sums: the loader thread of a column group adds up its columns.
template <int NBK, int PANEL>
__device__ __forceinline__ void niw_stats_body_shared(const StatsArgs &A, int seg, int cnt, double *__restrict__ slab, double *__restrict__ xb) {
    constexpr int SB = DPMM_STATS16_SB, NPT = 4 * SB;              // k-steps / points per batch
    constexpr int NCG = 4 * NBK;                                    // column groups (of four columns) per point
    static_assert(64 * StatCfg<NBK>::NPANEL == 8 * NCG, "eight point rows per loader pass");
    using C = StatCfg<NBK>;
    constexpr int P0 = PANEL * C::PP;
    constexpr int P1 = (P0 + C::PP < C::NPAIR) ? P0 + C::PP : C::NPAIR;
    constexpr int NP = P1 - P0;
    constexpr int BUF = NBK * NPT * 16;                             // doubles per LDS buffer
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 15, g = lane >> 4;
    // loader role: columns 4 cg .. 4 cg + 3 of the points prow, prow + 8, .. of a batch (threads = 8 point rows x NCG column groups)
    const int cg = tid % NCG, prow = tid / NCG;
    const bool colok = 4 * cg < A.ldx;
    const int coloff = colok ? 4 * cg : 0;
    const int li = (4 * cg) / NBK, lb = (4 * cg) % NBK;            // lane column and first block of the thread's four columns (lane i owns columns NBK i ..)
    f64x4 acc[NP > 0 ? NP : 1];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = (f64x4){0., 0., 0., 0.};
    double xs[4] = {0., 0., 0., 0.};
    const int nbatch = (cnt + NPT - 1) / NPT;
    constexpr int HL = NPT / 8;                                     // points per loader thread and batch (rows prow, prow + 8, ...)
    auto load_idx = [&](int bt, int (&pt)[HL]) {
#pragma unroll
        for (int h = 0; h < HL; ++h) pt[h] = A.sb.perm[seg + min(NPT * bt + prow + 8 * h, cnt - 1)];
    };
    auto load_x = [&](const int (&pt)[HL], f32x4 (&xv)[HL]) {
#pragma unroll
        for (int h = 0; h < HL; ++h) xv[h] = *reinterpret_cast<const f32x4 *>(A.X + (int64_t)pt[h] * A.ldx + coloff);
    };
    auto stage = [&](int bt, const f32x4 (&xv)[HL], double *dst) {   // convert + mask + store in operand order + column sums
#pragma unroll
        for (int h = 0; h < HL; ++h) {
            const int p = prow + 8 * h;
            const bool keep = colok && NPT * bt + p < cnt;
            const double d0 = keep ? (double)xv[h].x : 0.0, d1 = keep ? (double)xv[h].y : 0.0;
            const double d2 = keep ? (double)xv[h].z : 0.0, d3 = keep ? (double)xv[h].w : 0.0;
            xs[0] += d0; xs[1] += d1; xs[2] += d2; xs[3] += d3;
            double *q = dst + ((lb * NPT + p) * 16 + li);
            q[0] = d0; q[NPT * 16] = d1; q[2 * NPT * 16] = d2; q[3 * NPT * 16] = d3;
        }
    };
    int pt_a[HL], pt_b[HL];
    f32x4 xv[HL];
    load_idx(0, pt_a);
    load_x(pt_a, xv);
    load_idx(1, pt_a);
    stage(0, xv, xb);
    __syncthreads();
    for (int bt = 0; bt < nbatch; ++bt) {
        const double *cur = xb + (bt & 1) * BUF;
        load_idx(bt + 2, pt_b);                                      // (clamped: harmless beyond the item)
        __builtin_amdgcn_sched_barrier(0);
        load_x(pt_a, xv);                                            // batch bt + 1
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < SB; ++ks) {
            double xd[NBK];
#pragma unroll
            for (int b2 = 0; b2 < NBK; ++b2) xd[b2] = cur[(b2 * NPT + 4 * ks + g) * 16 + i];
#pragma unroll
            for (int ba = 0; ba < NBK; ++ba)
#pragma unroll
                for (int bb = 0; bb <= ba; ++bb) {
                    const int p = ba * (ba + 1) / 2 + bb;
                    if (p >= P0 && p < P1) acc[p - P0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[ba], xd[bb], acc[p - P0], 0, 0, 0);
                }
        }
        if (bt + 1 < nbatch) stage(bt + 1, xv, xb + ((bt + 1) & 1) * BUF);
#pragma unroll
        for (int h = 0; h < HL; ++h) pt_a[h] = pt_b[h];
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(int64_t)(P0 + p) * 256 + r * 64 + lane] = acc[p][r];
    // column sums: the four loader rows of a column group -> one value per column (fixed order: reproducible)
    constexpr int DPc = 16 * NBK;
    double *red = xb;                                                // [8][DP] (the buffers are free: the loop ended with a barrier)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[prow * DPc + 4 * cg + e] = xs[e];
    __syncthreads();
    if (tid < DPc)
        slab[(int64_t)C::NPAIR * 256 + tid] = ((red[tid] + red[DPc + tid]) + (red[2 * DPc + tid] + red[3 * DPc + tid])) +
                                               ((red[4 * DPc + tid] + red[5 * DPc + tid]) + (red[6 * DPc + tid] + red[7 * DPc + tid]));
    __syncthreads();
}

// Workgroup w owns the contiguous item range [B(w), B(w+1)), B(w) = floor(w T / G) -- T items over G = min(groups, T) workgroups, so
// that two ranges differ by at most ONE item.  (Round 3 cut ranges of q = ceil(T / groups) items: with the derived statistics a pass
// at the 8-GPU shard size has T = 2 200 items for 1 024 groups, q = 3, and 733 workgroups did all the work on a third of the chip
// while the rest had none: 0.124 ms where T / groups of the N = 1e7 pass predicts 0.05.)  Consecutive items of one bin are contiguous
// in perm, so the range splits into one segment per bin it touches; each segment is accumulated in registers and written as ONE
// slab, stored at the slot of its first item (its "head").  Heads of bin b are item_start[b] and every B(w) strictly inside the
// bin -- the reduce kernel walks exactly those (range_bound / range_heads below: both kernels use the same two functions).
__device__ __forceinline__ int range_bound(int w, int T, int G) { return (int)(((long long)w * T) / G); }
// boundaries B(w) with i0 < B(w) < i1: w in [wa, wb] (empty when wb < wa).  B(w) > i0 <=> w T >= (i0 + 1) G;  B(w) < i1 <=> w T < i1 G.
__device__ __forceinline__ void range_heads(int i0, int i1, int T, int G, int &wa, int &wb) {
    wa = (int)((((long long)i0 + 1) * G + T - 1) / T);
    wb = (int)(((long long)i1 * G + T - 1) / T) - 1;
}
// Slab SLOTS (the items used to name them: one 21 KB slab per item id, i.e. n / chunk of them allocated -- which tied the granularity of
// the balance to the memory the context holds): the head at a workgroup boundary B(w) lives in slot w, a head at the start of a bin that
// falls INSIDE a workgroup's range in slot NIW_STATS_MAX_GROUPS + bin.  NIW_STATS_MAX_GROUPS + 2 K slots whatever the item size.
__device__ __forceinline__ int range_owner(int item, int T, int G) { return (int)((((long long)item + 1) * G + T - 1) / T) - 1; }   // w with B(w) <= item < B(w+1)
__device__ __forceinline__ int head_slot(int item, int bin, int T, int G) {
    const int w = range_owner(item, T, G);
    return range_bound(w, T, G) == item ? w : NIW_STATS_MAX_GROUPS + bin;
}
template <int NBK>
__global__ __launch_bounds__(64 * StatCfg<NBK>::NPANEL) void niw_stats_kernel(StatsArgs A) {
    using C = StatCfg<NBK>;
    // The three small tables of the sort (item_start, bin_total, bin_start: nbins + 1 entries) in REGISTERS, two entries per lane, fetched
    // with one round trip: the total, the binary search of find_bin and the look-ups behind it were ~10 dependent L2 latencies in front of
    // a workgroup's first matrix instruction -- a fifth of the kernel at the 8-GPU shard size (550 points per workgroup).  nbins <= 127.
    const bool regtab = A.nbins < 128;
    const int tl = threadIdx.x & 63;
    int ts0 = 0, ts1 = 0, tt0 = 0, tt1 = 0, tb0 = 0, tb1 = 0;
    if (regtab) {
        const int e0 = min(tl, A.nbins), e1 = min(tl + 64, A.nbins);
        ts0 = A.sb.item_start[e0]; ts1 = A.sb.item_start[e1];
        tt0 = A.sb.bin_total[min(e0, A.nbins - 1)]; tt1 = A.sb.bin_total[min(e1, A.nbins - 1)];
        tb0 = A.sb.bin_start[e0]; tb1 = A.sb.bin_start[e1];
    }
    const int total_items = regtab ? (A.nbins < 64 ? __shfl(ts0, A.nbins) : __shfl(ts1, A.nbins - 64)) : A.sb.item_start[A.nbins];
    const int G = min((int)gridDim.x, total_items);
    if ((int)blockIdx.x >= G) return;
    const int it0 = range_bound((int)blockIdx.x, total_items, G);
    const int it1 = range_bound((int)blockIdx.x + 1, total_items, G);
    auto tab = [&](int v0, int v1, int idx) -> int { return idx < 64 ? __shfl(v0, idx) : __shfl(v1, idx - 64); };
    for (int item = it0; item < it1;) {
        int b, istart_b, istart_b1, bcnt, bstart;
        if (regtab) {
            // largest b < nbins with item_start[b] <= item (empty bins share a start with their successor: the LAST one wins, it holds the item)
            const unsigned long long m0 = __ballot(tl < A.nbins && ts0 <= item), m1 = __ballot(tl + 64 < A.nbins && ts1 <= item);
            b = m1 ? 127 - __clzll((long long)m1) : 63 - __clzll((long long)m0);
            istart_b = tab(ts0, ts1, b); istart_b1 = tab(ts0, ts1, b + 1);
            bcnt = tab(tt0, tt1, b); bstart = tab(tb0, tb1, b);
        } else {
            b = find_bin(A.sb.item_start, A.nbins, item);
            istart_b = A.sb.item_start[b]; istart_b1 = A.sb.item_start[b + 1];
            bcnt = A.sb.bin_total[b]; bstart = A.sb.bin_start[b];
        }
        const int e = min(it1, istart_b1);
        const int j = item - istart_b;
        const int seg = bstart + j * A.chunk;
        const int cnt = min((e - item) * A.chunk, bcnt - j * A.chunk);
        double *slab = A.slabs + (int64_t)(item == it0 ? (int)blockIdx.x : NIW_STATS_MAX_GROUPS + b) * A.slab_stride;
        const int panel = threadIdx.x >> 6;
        if constexpr (C::NPANEL == 1) {
            niw_stats_body<NBK, 0>(A, seg, cnt, slab);
        } else {
            __shared__ double xbs[2 * NBK * 4 * DPMM_STATS16_SB * 16];      // two operand buffers of niw_stats_body_shared
            if constexpr (C::NPANEL == 4) {
                switch (panel) {
                    case 0: niw_stats_body_shared<NBK, 0>(A, seg, cnt, slab, xbs); break;
                    case 1: niw_stats_body_shared<NBK, 1>(A, seg, cnt, slab, xbs); break;
                    case 2: niw_stats_body_shared<NBK, 2>(A, seg, cnt, slab, xbs); break;
                    default: niw_stats_body_shared<NBK, 3>(A, seg, cnt, slab, xbs); break;
                }
            } else {
                switch (panel) {
                    case 0: niw_stats_body_shared<NBK, 0>(A, seg, cnt, slab, xbs); break;
                    case 1: niw_stats_body_shared<NBK, 1>(A, seg, cnt, slab, xbs); break;
                    case 2: niw_stats_body_shared<NBK, 2>(A, seg, cnt, slab, xbs); break;
                    case 3: niw_stats_body_shared<NBK, 3>(A, seg, cnt, slab, xbs); break;
                    case 4: niw_stats_body_shared<NBK, 4>(A, seg, cnt, slab, xbs); break;
                    case 5: niw_stats_body_shared<NBK, 5>(A, seg, cnt, slab, xbs); break;
                    case 6: niw_stats_body_shared<NBK, 6>(A, seg, cnt, slab, xbs); break;
                    default: niw_stats_body_shared<NBK, 7>(A, seg, cnt, slab, xbs); break;
                }
            }
        }
        item = e;
    }
}

// position of packed-row element e (>= 1) inside a statistics slab (MFMA fragment order), computed once per context
__global__ void niw_row_offsets_kernel(int32_t *__restrict__ row_off, int D, int NBK, int64_t packed_stride) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= packed_stride) return;
    const int NPAIR = NBK * (NBK + 1) / 2;
    int64_t off = 0;
    if (e >= 1) {
        if (e <= D) {
            off = (int64_t)NPAIR * 256 + (e - 1);
        } else {
            const int64_t te = e - 1 - D;
            int a = (int)((sqrt(8.0 * (double)te + 1.0) - 1.0) * 0.5);
            while ((int64_t)(a + 1) * (a + 2) / 2 <= te) ++a;
            while ((int64_t)a * (a + 1) / 2 > te) --a;
            const int c = (int)(te - (int64_t)a * (a + 1) / 2);
            int ia = a / NBK, ba = a % NBK, ib = c / NBK, bb = c % NBK;
            int R, Cc, pa, pb;
            if (ba >= bb) { pa = ba; pb = bb; R = ia; Cc = ib; }
            else { pa = bb; pb = ba; R = ib; Cc = ia; }
            const int pair = pa * (pa + 1) / 2 + pb;
            const int lane = Cc + 16 * (R & 3);
            const int r = R >> 2;
            off = (int64_t)pair * 256 + r * 64 + lane;
        }
    }
    row_off[e] = (int32_t)off;
}
__global__ void niw_inv_offsets_kernel(const int32_t *__restrict__ row_off, int32_t *__restrict__ inv_off, int64_t packed_stride, int64_t slab_stride, int phase) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (phase == 0) { if (e < slab_stride) inv_off[e] = -1; }
    else if (e >= 1 && e < packed_stride) inv_off[row_off[e]] = (int32_t)e;
}
hipError_t launch_niw_row_offsets(int32_t *row_off, int32_t *inv_off, int D, int64_t packed_stride, hipStream_t s) {
    const int NBK = D <= 16 ? 1 : D <= 32 ? 2 : D <= 64 ? 4 : D <= 128 ? 8 : 16;
    const int64_t slab = niw_slab_stride_nbk(NBK);
    DPMM_LAUNCH(niw_row_offsets_kernel, dim3((unsigned)((packed_stride + 255) / 256)), dim3(256), 0, s, row_off, D, NBK, packed_stride);
    DPMM_LAUNCH(niw_inv_offsets_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, s, row_off, inv_off, packed_stride, slab, 0);
    DPMM_LAUNCH(niw_inv_offsets_kernel, dim3((unsigned)((packed_stride + 255) / 256)), dim3(256), 0, s, row_off, inv_off, packed_stride, slab, 1);
    return hipGetLastError();
}

// packed row: [0] N, [1..D] sum, [1+D + a(a+1)/2 + b] S[a][b] (a >= b)
// block = 64 row elements x REDUCE_PARTS parts: a bin's segment heads (one slab each, <= range_groups + 1 of them, many more for a
// large cluster than for a small one) are cut into REDUCE_PARTS contiguous runs summed by different threads, eight loads in flight
// each, and the partial sums are combined in part order -- a fixed summation tree, so the rows stay bitwise reproducible.
constexpr int REDUCE_PARTS = 4;
// One workgroup = 64 slab positions of ONE CLUSTER (both of its bins, one after the other: in a derived pass only one of them was
// computed).  With `A.mode` (per-step pass with derived statistics) the derivation of the sub-cluster that was NOT computed happens right
// here, on the sums the thread just formed, instead of in a launch of its own behind this one (derive_rows_kernel: still used by the
// Multinomial path): mode[k] = 0: cache[k] = left + right; 1 / 2: left / right = cache[k] - the computed row.  N comes from the bin totals
// either way (integers, known for every bin whether or not it was selected).  Same heads, same parts, same order per element as before:
// bitwise the same rows.
__global__ __launch_bounds__(64 * REDUCE_PARTS) void niw_reduce_kernel(StatsArgs A, int NBK) {
    // a thread owns a slab POSITION (consecutive threads read consecutive doubles of a slab: the gather through row_off touched runs of
    // at most 16) and writes its sum to the packed-row element the inverse table names
    __shared__ double part_sum[2][REDUCE_PARTS][64];
    const int k = blockIdx.y;
    const int el = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t pos = blockIdx.x * 64ll + el;
    const int e = pos < A.slab_stride ? A.inv_off[pos] : -1;           // packed-row element of this position (>= 1) or none
    const bool live = e >= 1;
    const int total_items = A.sb.item_start[A.nbins];
    const int G = min(A.range_groups, total_items);                          // as in niw_stats_kernel
    bool sel[2];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
        const int b = 2 * k + sd;
        sel[sd] = A.sb.bin_sel[b] != 0;
        double s = 0.;
        if (sel[sd] && live) {
            const int i0 = A.sb.item_start[b], i1 = A.sb.item_start[b + 1];
            // segment heads of the bin, in item order: i0, then every workgroup boundary strictly inside (i0, i1)
            int wa = 1, wb = 0;
            if (i1 > i0) range_heads(i0, i1, total_items, G, wa, wb);
            const int nheads = i1 > i0 ? 1 + max(0, wb - wa + 1) : 0;
            const int slot0 = i1 > i0 ? head_slot(i0, b, total_items, G) : 0;
            const int h0 = (int)((int64_t)nheads * part / REDUCE_PARTS), h1 = (int)((int64_t)nheads * (part + 1) / REDUCE_PARTS);
            for (int h = h0; h < h1; h += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int hh = h + u;
                    const int slot = hh == 0 ? slot0 : wa + hh - 1;
                    v[u] = hh < h1 ? A.slabs[(int64_t)slot * A.slab_stride + pos] : 0.;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (h + u < h1) s += v[u];
            }
        }
        part_sum[sd][part][el] = s;
    }
    __syncthreads();
    if (part != 0) return;
    double *outl = A.out + (int64_t)(2 * k) * A.packed_stride, *outr = outl + A.packed_stride;
    const int m = A.mode ? (int)A.mode[k] : -1;
    double *c = A.mode ? A.cache + (int64_t)k * A.packed_stride : nullptr;
    if (pos == 0) {
        // element 0 = N: the bin totals, exact for computed and derived rows alike
        double nl = (double)A.sb.bin_total[2 * k], nr = (double)A.sb.bin_total[2 * k + 1];
        if (m == 0) c[0] = nl + nr;
        if (A.cside) {
            const int cs = A.cside[k];
            A.out[(int64_t)(A.nbins + k) * A.packed_stride] = cs ? nl : 0.;
            if (cs) { const double t = nl + nr; nl = cs == 1 ? t : 0.; nr = cs == 2 ? t : 0.; }
            if (k == 0) { A.zero2[0] = 0; A.zero2[1] = 0; }          // flags[K] (any bad), flags[K + 1] (some candidate was not bad): set by the finalize kernel
        }
        outl[0] = (m >= 0 || sel[0]) ? nl : 0.;
        outr[0] = (m >= 0 || sel[1]) ? nr : 0.;
        if (A.mode) {
            A.dirty[k] = 0;
            if (k == 0) A.dirty[DPMM_MAX_CLUSTERS_K] = 0;
        }
    }
    // rider: the bad-cluster flags of the pass go to the host's pinned block from here (no copy kernel of their own)
    if (A.flags_dst && blockIdx.x == 0 && k == 0) for (int i = el; i <= A.K; i += 64) A.flags_dst[i] = A.flags_src[i];
    if (!live) return;
    double sl = part_sum[0][0][el], sr = part_sum[1][0][el];
#pragma unroll
    for (int p = 1; p < REDUCE_PARTS; ++p) { sl += part_sum[0][p][el]; sr += part_sum[1][p][el]; }
    if (m == 0) c[e] = sl + sr;
    else if (m == 1) sl = c[e] - sr;
    else if (m == 2) sr = c[e] - sl;
    if (A.cside) {
        // one-collective pass: sl / sr belong to the SPECULATIVELY reset sub-labels of a candidate (cside[k] != 0).  What travels is (a) the
        // rows of the labels as swept -- everything on the side the shard's points were on -- and (b) the re-drawn left row X
        const int cs = A.cside[k];
        A.out[(int64_t)(A.nbins + k) * A.packed_stride + e] = cs ? sl : 0.;
        if (cs) { const double t = sl + sr; sl = cs == 1 ? t : 0.; sr = cs == 2 ? t : 0.; }
    }
    outl[e] = sl;                                                        // (a bin that was neither selected nor derived: zero)
    outr[e] = sr;
}

// One-collective per-step pass, behind the all-reduce of `red` = [2K rows of the labels as swept | K re-drawn left rows X] (all summed over
// the ranks): reset_bad_clusters! (src/local_clusters_actions.jl:501-516) decided from the GLOBAL occupancies in the rows' own N column.
// Cluster k is bad when N_l = 0 or N_r = 0; then EVERY rank holding points of it had it as a candidate, applied the reset and added its
// part of X: left' = X, right' = (left + right) - X (one side of a bad cluster is empty: left + right is its cluster-level row).  Otherwise
// the rows are taken as they are, and a shard on which k was a candidate undoes its reset (flags[K + 1], niw_undo_reset_kernel).
// Reads `red`, writes `out` + flags (flags[K], flags[K + 1] were cleared by the reduce kernel).
__global__ __launch_bounds__(256) void niw_finalize_rows_kernel(const double *__restrict__ red, double *__restrict__ out, int64_t stride, int K,
                                                                const uint8_t *__restrict__ cside, uint8_t *__restrict__ flags, uint8_t *__restrict__ flags_host) {
    const int k = blockIdx.y;
    const int64_t e = blockIdx.x * 256ll + threadIdx.x;
    const double *l = red + (int64_t)(2 * k) * stride, *r = l + stride;
    const bool bad = l[0] == 0. || r[0] == 0.;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        flags[k] = bad ? 1 : 0;
        if (bad) flags[K] = 1;
        if (!bad && cside[k]) flags[K + 1] = 1;
        // rider: the verdict goes to the caller's pinned block from here (flags_host[K] = "any" was cleared by the host before the launch)
        if (flags_host) { flags_host[k] = bad ? 1 : 0; if (bad) flags_host[K] = 1; }
    }
    if (e >= stride) return;
    double *ol = out + (int64_t)(2 * k) * stride, *orr = ol + stride;
    if (bad) {
        const double x = red[(int64_t)(2 * K + k) * stride + e];
        ol[e] = x;
        orr[e] = (l[e] + r[e]) - x;
    } else {
        ol[e] = l[e];
        orr[e] = r[e];
    }
}
// One-collective per-step pass of a prior whose reduce does not write the travelling rows itself (Multinomial): rows [2K][stride] of the
// SPECULATIVELY reset labels (computed or derived) -> red = [2K rows of the labels as swept | K re-drawn left rows X], exactly what
// niw_reduce_kernel emits for the NIW prior: for a candidate (cside[k] = 1 / 2: all of the shard's points of k were on that side) the side
// its points were on carries left' + right', the other one zeros, X = left'; for every other cluster the rows as they are and X = 0.
// flags[K], flags[K + 1] are cleared for the finalize kernel.
__global__ __launch_bounds__(256) void onecoll_rows_kernel(const double *__restrict__ rows, double *__restrict__ red, int64_t stride, int K,
                                                           const uint8_t *__restrict__ cside, uint8_t *__restrict__ flags) {
    const int k = blockIdx.y;
    const int64_t e = blockIdx.x * 256ll + threadIdx.x;
    if (k == 0 && blockIdx.x == 0 && threadIdx.x == 0) { flags[K] = 0; flags[K + 1] = 0; }
    if (e >= stride) return;
    double sl = rows[(int64_t)(2 * k) * stride + e], sr = rows[(int64_t)(2 * k + 1) * stride + e];
    const int cs = cside[k];
    red[(int64_t)(2 * K + k) * stride + e] = cs ? sl : 0.;
    if (cs) { const double t = sl + sr; sl = cs == 1 ? t : 0.; sr = cs == 2 ? t : 0.; }
    red[(int64_t)(2 * k) * stride + e] = sl;
    red[(int64_t)(2 * k + 1) * stride + e] = sr;
}
hipError_t launch_onecoll_rows(const double *rows, double *red, int64_t stride, int K, const uint8_t *cside, uint8_t *flags, hipStream_t s) {
    DPMM_LAUNCH(onecoll_rows_kernel, dim3((unsigned)((stride + 255) / 256), K), dim3(256), 0, s, rows, red, stride, K, cside, flags);
    return hipGetLastError();
}
hipError_t launch_niw_finalize_rows(const double *red, double *out, int64_t stride, int K, const uint8_t *cside, uint8_t *flags, uint8_t *flags_host, hipStream_t s) {
    DPMM_LAUNCH(niw_finalize_rows_kernel, dim3((unsigned)((stride + 255) / 256), K), dim3(256), 0, s, red, out, stride, K, cside, flags, flags_host);
    return hipGetLastError();
}
// A candidate of this shard that is not bad globally: all of its points go back to the side they were on (cside: 1 left, 2 right).
__global__ void niw_undo_reset_kernel(int32_t *__restrict__ bins, int64_t n, int K, const uint8_t *__restrict__ flags, const uint8_t *__restrict__ cside) {
    if (!flags[K + 1]) return;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = bins[i];
        const int z = b >> 1;
        if (b >= 0 && z < K && cside[z] && !flags[z]) bins[i] = 2 * z + (cside[z] - 1);
    }
}
hipError_t launch_niw_undo_reset(int32_t *bins, int64_t n, int K, const uint8_t *flags, const uint8_t *cside, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    DPMM_LAUNCH(niw_undo_reset_kernel, dim3(grid), dim3(256), 0, s, bins, n, K, flags, cside);
    return hipGetLastError();
}

int64_t niw_slab_stride(int D) {
    const int NBK = D <= 16 ? 1 : D <= 32 ? 2 : D <= 64 ? 4 : D <= 128 ? 8 : 16;
    return niw_slab_stride_nbk(NBK);
}

static int niw_nbk(int D) { return D <= 16 ? 1 : D <= 32 ? 2 : D <= 64 ? 4 : D <= 128 ? 8 : 16; }

hipError_t launch_niw_stats(const StatsArgs &a0, hipStream_t s) {
    StatsArgs a = a0;
    const int NBK = niw_nbk(a.D);
    // two resident waves per SIMD at D = 64 (register budget): 2048 workgroups cover the chip once; ranges are balanced to one item, so
    // nothing is gained from a longer grid.  Measured (scripts/stats_groups_sweep.py): D <= 64: 2048 at N = 1e7 (0.98 ms; 1024: 1.00,
    // 512: 1.84); D = 128 (four panels on the shared data path): 512 (0.50 ms; 256: 0.55, 1024: 0.53, 2048: 0.60); D = 256 (one
    // workgroup per compute unit fits): 256.  Small passes: a workgroup per ~128 points at least (a slab is 21 KB at D = 64 -- as much
    // as 80 points of input -- and every one is written, read back and summed by the reduce: 1024 groups below 2.5e6 points, where the
    // reduce of 2048 slabs costs more than the second wave per SIMD gains).
    int dflt = NBK <= 4 ? (a.n >= 2500000 ? 2048 : 1024) : (NBK <= 8 ? 512 : 256);
    if (NBK <= 4) { const int64_t by_points = (a.n + 127) / 128; if (by_points < dflt) dflt = (int)(by_points < 64 ? 64 : by_points); }
    int groups = a.range_groups > 0 ? a.range_groups : dflt;
    if (groups > NIW_STATS_MAX_GROUPS) groups = NIW_STATS_MAX_GROUPS;
    a.range_groups = groups;
    switch (NBK) {
        case 1: DPMM_LAUNCH((niw_stats_kernel<1>), dim3(groups), dim3(64), 0, s, a); break;
        case 2: DPMM_LAUNCH((niw_stats_kernel<2>), dim3(groups), dim3(64), 0, s, a); break;
        case 4: DPMM_LAUNCH((niw_stats_kernel<4>), dim3(groups), dim3(64), 0, s, a); break;
        case 8: DPMM_LAUNCH((niw_stats_kernel<8>), dim3(groups), dim3(256), 0, s, a); break;
        default: DPMM_LAUNCH((niw_stats_kernel<16>), dim3(groups), dim3(512), 0, s, a); break;
    }
    DPMM_LAUNCH(niw_reduce_kernel, dim3((unsigned)((a.slab_stride + 63) / 64), a.nbins / 2), dim3(64 * REDUCE_PARTS), 0, s, a, NBK);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ Multinomial statistics
// item = (bin, <= chunk points); 256 threads, thread t owns dims t, t+256, ...; Float64 sums
// (exact for count data; rounded to Float32 by the host as the reference stores them).
__global__ __launch_bounds__(256) void mult_stats_kernel(StatsArgs A) {
    const int total_items = A.sb.item_start[A.nbins];
    for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
        const int b = find_bin(A.sb.item_start, A.nbins, item);
        const int j = item - A.sb.item_start[b];
        const int bcnt = A.sb.bin_total[b];
        const int seg = A.sb.bin_start[b] + j * A.chunk;
        const int cnt = min(A.chunk, bcnt - j * A.chunk);
        double *slab = A.slabs + (int64_t)item * A.slab_stride;
        for (int d0 = 0; d0 < A.D; d0 += 256 * 4) {
            double s[4] = {0., 0., 0., 0.};
            int dq[4];
            bool okq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int d = d0 + q * 256 + threadIdx.x; okq[q] = d < A.D; dq[q] = okq[q] ? d : 0; }
            // 8 points per trip, all 32 loads issued before the first add (unconditional, clamped column): the sums stay
            // in point order, so the result is bit-identical to the one-point-at-a-time loop
            int p = 0;
            for (; p + 8 <= cnt; p += 8) {
                float v[8][4];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float *xp = A.X + (int64_t)A.sb.perm[seg + p + u] * A.ldx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[u][q] = xp[dq[q]];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) s[q] += (double)v[u][q];
            }
            for (; p < cnt; ++p) {
                const float *xp = A.X + (int64_t)A.sb.perm[seg + p] * A.ldx;
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] += (double)xp[dq[q]];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (!okq[q]) s[q] = 0.;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = d0 + q * 256 + threadIdx.x;
                if (d < A.D) slab[d] = s[q];
            }
        }
    }
}

__global__ __launch_bounds__(256) void mult_reduce_kernel(StatsArgs A) {
    const int b = blockIdx.y;
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= A.packed_stride) return;
    double *out = A.out + (int64_t)b * A.packed_stride;
    if (!A.sb.bin_sel[b]) { out[e] = 0.; return; }
    if (e == 0) { out[0] = (double)A.sb.bin_total[b]; return; }
    double s = 0.;
    const int i0 = A.sb.item_start[b], i1 = A.sb.item_start[b + 1];
    // eight loads in flight, added in item order (the same sum as one at a time: a bin has 30 - 60 items at N = 1e6, each a trip to L2)
    int it = i0;
    for (; it + 8 <= i1; it += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = A.slabs[(int64_t)(it + u) * A.slab_stride + (e - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; it < i1; ++it) s += A.slabs[(int64_t)it * A.slab_stride + (e - 1)];
    out[e] = s;
}

// u8 path: the same segmented column sums over the byte copy of the points.  Thread t owns the four dimensions of dword t of a
// 1024-byte block of the row; the counts are added as INTEGERS (exact; a work item holds far fewer than 2^24 points) and leave
// as Float64 slabs, so the reduce kernel and the packed rows are unchanged -- and bit-identical to the Float32-reading kernel.
__global__ __launch_bounds__(256) void mult_stats_u8_kernel(StatsArgs A, const uint8_t *__restrict__ X8, int64_t ld8) {
    const int total_items = A.sb.item_start[A.nbins];
    for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
        const int b = find_bin(A.sb.item_start, A.nbins, item);
        const int j = item - A.sb.item_start[b];
        const int bcnt = A.sb.bin_total[b];
        const int seg = A.sb.bin_start[b] + j * A.chunk;
        const int cnt = min(A.chunk, bcnt - j * A.chunk);
        double *slab = A.slabs + (int64_t)item * A.slab_stride;
        for (int64_t d0 = 0; d0 < ld8; d0 += 1024) {
            const bool ok = d0 + 4 * threadIdx.x < ld8;
            const int64_t off = ok ? d0 + 4 * (int64_t)threadIdx.x : 0;
            uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int p = 0;
            for (; p + 8 <= cnt; p += 8) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint32_t *>(X8 + (int64_t)A.sb.perm[seg + p + u] * ld8 + off);
#pragma unroll
                for (int u = 0; u < 8; ++u) { s0 += v[u] & 0xffu; s1 += (v[u] >> 8) & 0xffu; s2 += (v[u] >> 16) & 0xffu; s3 += v[u] >> 24; }
            }
            for (; p < cnt; ++p) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(X8 + (int64_t)A.sb.perm[seg + p] * ld8 + off);
                s0 += v & 0xffu; s1 += (v >> 8) & 0xffu; s2 += (v >> 16) & 0xffu; s3 += v >> 24;
            }
            if (ok) {
                const int64_t d = d0 + 4 * (int64_t)threadIdx.x;
                if (d < A.D) slab[d] = (double)s0;
                if (d + 1 < A.D) slab[d + 1] = (double)s1;
                if (d + 2 < A.D) slab[d + 2] = (double)s2;
                if (d + 3 < A.D) slab[d + 3] = (double)s3;
            }
        }
    }
}

hipError_t launch_mult_stats_u8(const StatsArgs &a, const uint8_t *X8, int64_t ld8, hipStream_t s) {
    const int grid = a.max_items < 1 ? 1 : a.max_items;
    DPMM_LAUNCH(mult_stats_u8_kernel, dim3(grid), dim3(256), 0, s, a, X8, ld8);
    DPMM_LAUNCH(mult_reduce_kernel, dim3((unsigned)((a.packed_stride + 255) / 256), a.nbins), dim3(256), 0, s, a);
    return hipGetLastError();
}

int64_t mult_slab_stride(int D) { return D; }

hipError_t launch_mult_stats(const StatsArgs &a, hipStream_t s) {
    const int grid = a.max_items < 1 ? 1 : a.max_items;
    DPMM_LAUNCH(mult_stats_kernel, dim3(grid), dim3(256), 0, s, a);
    DPMM_LAUNCH(mult_reduce_kernel, dim3((unsigned)((a.packed_stride + 255) / 256), a.nbins), dim3(256), 0, s, a);
    return hipGetLastError();
}

// second half of the sort (needs the selection mask and the chunk size of the statistics pass)
static void launch_scatter(const int32_t *bins, const StatsArgs &a, int nt, hipStream_t s) {
    const StepStarts none{};
    int32_t *b = const_cast<int32_t *>(bins);          // (only the STEP instantiation with reset_flags writes labels)
    DPMM_TILE_DISPATCH(a.sb.tile,
        DPMM_LAUNCH((scatter_kernel<512, false>), dim3(nt), dim3(64), a.nbins * sizeof(int), s, b, a.n, a.nbins, nt, a.sb.tile_hist, a.sb.bin_start, a.sb.perm, none),
        DPMM_LAUNCH((scatter_kernel<2048, false>), dim3(nt), dim3(64), a.nbins * sizeof(int), s, b, a.n, a.nbins, nt, a.sb.tile_hist, a.sb.bin_start, a.sb.perm, none));
}
// rs (nullable): the bad-cluster reset folded into this chain -- the histogram counted it ahead (launch_step_hist with spec), the scan derives the
// flags and picks the counts, the scatter applies it (needs fused_starts: the STEP instantiation)
hipError_t launch_step_scan_scatter(int32_t *bins, const StatsArgs &a, int derive, int force_all, int fused_starts, const StepReset *rs, hipStream_t s) {
    const int nt = sort_nt(a.n, a.sb);
    if (rs) DPMM_LAUNCH(scan_tiles_step_kernel, dim3(a.nbins), dim3(256), 0, s, a.sb.tile_cnt, a.sb.tile_spec, a.sb.tile_hist, nt, a.sb.bin_total, a.sb.fast_total,
                        rs->global_counts, rs->flags, rs->K, rs->cside);
    else DPMM_LAUNCH(scan_tiles_kernel, dim3(a.nbins), dim3(256), 0, s, a.sb.tile_cnt, a.sb.tile_hist, nt, a.sb.bin_total);
    if ((fused_starts || rs) && nt > 0) {          // the starts inside the scatter launch (StepStarts above)
        const StepStarts st{a.sb.bin_total, a.sb.bin_sel, a.chunk, a.sb.bin_start, a.sb.item_start, a.sb.perm_total, a.sb.fast_total,
                            derive ? a.sb.cmode : (uint8_t *)nullptr, a.sb.cdirty, force_all,
                            rs ? rs->flags : (const uint8_t *)nullptr, rs ? a.sb.spec_bins : (const int32_t *)nullptr, a.sb.tile_cnt};
        const size_t lds = a.nbins * sizeof(int) + (size_t)(((a.nbins >> 1) + 3) & ~3);
        DPMM_TILE_DISPATCH(a.sb.tile,
            DPMM_LAUNCH((scatter_kernel<512, true>), dim3(nt), dim3(64), lds, s, bins, a.n, a.nbins, nt, a.sb.tile_hist, a.sb.bin_start, a.sb.perm, st),
            DPMM_LAUNCH((scatter_kernel<2048, true>), dim3(nt), dim3(64), lds, s, bins, a.n, a.nbins, nt, a.sb.tile_hist, a.sb.bin_start, a.sb.perm, st));
        return hipGetLastError();
    }
    DPMM_LAUNCH(starts_step_kernel, dim3(1), dim3(256), 0, s, a.sb.bin_total, a.sb.bin_sel, a.nbins, a.chunk, a.sb.bin_start, a.sb.item_start,
                a.sb.perm_total, a.sb.fast_total, derive ? a.sb.cmode : (uint8_t *)nullptr, a.sb.cdirty, force_all);
    launch_scatter(bins, a, nt, s);
    return hipGetLastError();
}
hipError_t launch_sort_finish(const int32_t *bins, const StatsArgs &a, hipStream_t s) {
    const int nt = sort_nt(a.n, a.sb);
    DPMM_LAUNCH(starts_kernel, dim3(1), dim3(256), 0, s, a.sb.bin_total, a.sb.bin_sel, a.nbins, a.chunk,
                       a.sb.bin_start, a.sb.item_start, a.sb.perm_total);
    if (nt > 0) launch_scatter(bins, a, nt, s);
    return hipGetLastError();
}

}  // namespace dpmm
