// dpmm_kernels.h -- kernel argument blocks and host-side launchers (internal to libdpmmhip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DPMM_MAX_CLUSTERS_K 1024

namespace dpmm {

// Device-resident label state: bins[i] = 2*(label-1) + (sub_label-1)  (Int32).
// The (label, sub-label) pair of the reference (src/ds.jl:54-55) is exactly the
// sufficient-statistics bin the point contributes to.

constexpr int DPMM_WORK_QUEUES = 8;                               // queue heads of the D <= 64 sweep kernel, 16 u64 apart from work[8]
constexpr int DPMM_WORK_SLOTS = 8 + 16 * DPMM_WORK_QUEUES;        // first per-wave counter slot
constexpr int DPMM_WORK_PER_WAVE = 8;                             // u64 per wave slot (one 64-byte half line): wave tiles, full evaluations, 16-row screens, tail pairs, reference brackets, 3 spare
constexpr int REFB_FRAGS = 6, REFB_WORDS = REFB_FRAGS * 256;      // dwords of a cluster's bf16 image for the reference bracket (refb_map, dpmm_device.h)
constexpr int B3_WORDS = 3 * REFB_WORDS;                           // dwords of a matrix's three-plane bf16 image (planes h | m | l, each in the bracket's fragment layout): niw_lean.hip
constexpr int B3_DVEC = 64;                                        // floats of a sub-cluster's offset vector d = R_s (mu_k - mu_s)

struct NiwSweepArgs {
    const float *X;      // [n][ldx] points (zero padded to ldx = roundup(D,4))
    int64_t ldx;
    int64_t n;
    int64_t first_index; // global index of point 0 (RNG counter)
    int64_t ntiles;
    int K;
    const float *Rp;     // packed factor fragments [3K][NP][64][4]
    const float *mup;    // [3K][DP]
    const float *cst;    // [3K]: 3k: -logdet/2 + log w_k ; 3k+1+s: -logdet/2 + log lr_w[k][s]
    const float *tdf;    // null: Gaussian a = cst - q/2 ; else Student-t (posterior predictive): [3K][2] = {df, (df+D)/2}, a = cst - hdf*log1p(q/df)
    float *scratch;      // a_k rows: scratch[k*scratch_stride + base + point_in_tile]
    int64_t scratch_stride;
    int scratch_by_tile; // 1: base = tile*TILE (full table, debug) ; 0: base = blockIdx*TILE
    int labels_only;     // 1: stop after the label phase (debug_loglik)
    int32_t *bins;       // out
    const int32_t *order_total; // device scalar: number of points `order` covers (must equal n, else identity is used)
    const int32_t *order; // processing order (permutation of [0,n), e.g. sorted by the previous bins) or null = identity
    uint64_t seed;
    uint32_t epoch;
    int final_argmax;
    float screen_margin;      // > 0: skip clusters whose a_k is provably below (reference - margin) for a whole wave (NIW, D in 17..64)
    const float *tail;        // [ceil(K/2)][16][2] tail-screen records of the cluster-level matrices, pairs interleaved, then [K][16] ball records (null: no tail screen)
    int tail_g;               // row group (lane >> 4) whose x registers of the last block hold features D-4..D-1
    int ball;                 // 1: cluster-per-lane ball test in front of the per-point tail screen (records [K][16] behind the pair records)
    int bf16scr;              // bit 0: D in 33..64 with tail records: bf16 screens in front of the Float32 16-row screen and of every survivor's first row block (DPMM_OPT_BF16_SCREENS);
                              // bit 1 (with sp_frag): the direction screen runs in FRONT of the tail-pair tests (the sweeps between two measuring ones).
                              // (A field of its own for bit 1 cost the common kernel 32 spilled registers: the argument block's size decides how the compiler lays out its scalars.)
    int bracket;              // 1: D in 33..64, homogeneous waves: certified bf16 bracket of the reference cluster's value first; its Float32 evaluation only if a cluster survives the screens (bf16 images behind the ball records)
    const float *lam;         // [K] lower bounds of lambda_min(Sigma_k^-1) (null: no scalar pre-screen)
    const float *mdist;       // [K][K] distances between the cluster means
    int screen_lds;           // set by the launcher: screen operands of all K clusters are staged in LDS
    int use_prev;             // bins hold labels from a previous sweep (reference clusters of the screen)
    int lds_rows;             // set by the launcher: rows of the a_k table that live in LDS (0: global scratch)
    const uint32_t *sp_frag;  // direction screen (D in 33..64, K <= 64; null: none): [K][4][2][64][4] bf16 fragments of the pair directions w, k0 major (launch_niw_direction);
                              // D = 128, 256: one word per 128-point tile, 1 + k0 where launch_niw_bracket_big bracketed the tile's reference cluster, else 0 (null: none)
    const float *sp_cons;     // [K][3][64]: per reference cluster k0 the constants {b, e, cst} of every cluster (direction_far, niw_sweep.hip);
                              // D = 128, 256: per position of the visiting order the bracket's lower end of a_k0 (launch_niw_bracket_big)
    uint32_t *need;           // [waves][2] (pinned host memory, may be null): [0] = (candidates this wave's tiles kept behind the 4-row tests, saturating) << 16 | tiles
                              // (15 bits), bit 15: counted in FRONT of the tail-pair tests (direction screen first): an upper bound;
                              // [1] (written by the DIR kernel only) = (candidates its direction screens removed) << 16 | candidates they were given
    unsigned long long *dbg;  // diagnostic builds only (DPMM_STAMPS): per-wave phase cycle sums
    int queue_rounds;         // D <= 64 kernel: rounds of tiles handed out through the queue at the end of the launch (-1: automatic)
    int prio;                 // 1: s_setprio -- low while the wave streams MFMAs, high in its scalar / VALU phases (DPMM_OPT_WAVE_PRIO)
    unsigned long long *work; // [4] tile queue head of the LDS-staged kernel, [8 + 16 q], q < 8: the eight queue heads of the D <= 64 kernel (one 128-byte
                              // line each), all cleared before the launch; [DPMM_WORK_SLOTS + DPMM_WORK_PER_WAVE w ..]: executed-work counters of
                              // wave w of this launch (wave tiles, full evaluations, 16-row screens, tail-screened cluster pairs, reference brackets), plain
                              // stores at kernel end, summed by the reader; may be null
};

int niw_tile_points(int NB);
int niw_occupancy(int NB);  // resident 256-thread workgroups per CU the sweep kernel is built for
hipError_t launch_niw_sweep(int NB, const NiwSweepArgs &a, int grid, hipStream_t s);
hipError_t launch_niw_refb_debug(const NiwSweepArgs &a, int k, float c_override, float *qhi_out, float *q_out, hipStream_t s);   // needs X, ldx, n, K, Rp, mup, tail
hipError_t launch_niw_screen_prep(const float *R, const float *mu, int D, int K, float *lam, float *dist, const int32_t *slot, hipStream_t s);
hipError_t launch_niw_pack(const float *R, const float *mu, float *Rp, float *mup, int D, int NB, int nmat, float *tail, const float *cst,
                           const int32_t *slot, float *cst_out, unsigned long long *work, hipStream_t s);
// Tables of the direction screen from the sweep's own images (Rp, mup, cst of the K cluster-level distributions; D in 33..64, K <= SP_MAXK):
// frag [K][SP_FRAG_WORDS], cons [K][SP_CONS_FLOATS]
constexpr int SP_MAXK = 64, SP_FRAG_WORDS = 4 * 2 * 256, SP_CONS_FLOATS = 3 * 64;
// bf16 images of the K cluster-level factors for the reference bracket of the LDS-staged kernels (NB = 8, 16), from the Float32 fragment
// image: out [K][niw_refb_big_words(NB)]  (passed to the sweep as NiwSweepArgs::sp_frag)
inline size_t niw_refb_big_words(int NB) { size_t c = 0; for (int bi = 0; bi < NB; ++bi) c += NB / 2 - bi / 2; return c * 256; }
hipError_t launch_niw_refb_big(const float *Rp, int NB, int K, uint32_t *out, hipStream_t s);
// the bracket itself, in front of the sweep launch (same NiwSweepArgs: X, order, bins, mup, cst): tile_flag [ceil(n / 128)], aref [128 ceil(n / 128)]
hipError_t launch_niw_bracket_big(int NB, const NiwSweepArgs &a, const uint32_t *refb, uint32_t *tile_flag, float *aref, hipStream_t s);
// ---- niw_lean.hip: the bf16 three-plane evaluation of the sub-cluster quadratic forms (NB = 4)
// images [3K][B3_WORDS] and offset vectors [3K][B3_DVEC] of the 2K sub-cluster factors from the Float32 fragment image, written behind the bracket's images in `tail`
hipError_t launch_niw_b3_pack(const float *Rp, const float *mup, int K, float *tail, int what, hipStream_t s);      // what bit 0: images + offsets; bit 1: the pair-ball table (2 <= K <= PB_MAXK) behind them
// the pair-ball table behind the offsets (round 6; K <= PB_MAXK): pd [K][K] -- pd[k K + j] = a certified LOWER bound of |R_j (mu_k - mu_j)|, the distance of
// cluster k's mean from cluster j's in j's own metric -- and sn [K], certified UPPER bounds of the spectral norms |R_j|_2 (niw_pair_ball_kernel, niw_lean.hip)
constexpr int PB_MAXK = 256;
inline size_t niw_pair_ball_floats(size_t K) { return K <= (size_t)PB_MAXK ? K * K + K : 0; }
inline size_t niw_pair_ball_offset(size_t K) { return 32 * ((K + 1) >> 1) + 16 * K + (size_t)REFB_WORDS * K + 3 * K * (size_t)B3_WORDS + 3 * K * (size_t)B3_DVEC; }      // floats from `tail` (= b3_offsets(tail, K) + 3 K B3_DVEC, niw_b3.h)
inline size_t niw_tail_floats(size_t cap) {      // pair records | ball records | bracket images | b3 images | b3 offsets | pair-ball table
    return 16 * (cap + 2) + 16 * cap + (size_t)REFB_WORDS * cap + 3 * cap * (size_t)B3_WORDS + 3 * cap * (size_t)B3_DVEC + niw_pair_ball_floats(cap < (size_t)PB_MAXK ? cap : (size_t)PB_MAXK);
}
// out [2K][n]: both sub-cluster values of every point under every cluster (needs X, ldx, n, K, mup, cst, tail)
hipError_t launch_niw_b3_debug(const NiwSweepArgs &a, float *out, hipStream_t s);
// the sub-label phase alone for the wave tiles of `list` (list[0] = count, list[1 ..] = tile indices; null: all tiles); labels are read from bins
hipError_t launch_niw_sub(const NiwSweepArgs &a, const uint32_t *list, uint32_t *count_out, int grid, hipStream_t s);      // count_out (pinned, nullable): receives list[0]
// whole tiles where bracket + ball + tail screens settle them (every point had label k0, nothing else can compete): labels AND sub-labels; every
// other tile is appended to list ([0] = count, cleared by the caller; [1 ..] = wave-tile indices) and left untouched.  need2 [4 grid] (nullable, pinned): tiles settled per wave
constexpr int NIW_LEAN_MAX_BINS = 256;      // bins of the sort the lean kernel can align its tiles to (beyond: tiles of 64 consecutive positions)
// bin_start [nbins + 1] (nullable): the offsets of the sort that wrote a.order -- tiles never cross a bin.  The list holds (position, count) pairs:
// 1 + 2 * (ceil(n / 64) + nbins) words at most.
hipError_t launch_niw_lean(const NiwSweepArgs &a, uint32_t *list, uint32_t *need2, uint32_t *other_list, const int32_t *bin_start, int nbins, uint32_t *need3, int grid, hipStream_t s);      // list[0] must be 0; other_list[0] is cleared for the next launch
hipError_t launch_niw_direction(const float *Rp, const float *mup, const float *cst, int D, int K, uint32_t *frag, float *cons, hipStream_t s);

struct MultSweepArgs {
    const float *X;
    int64_t ldx;
    int64_t n;
    int64_t first_index;
    int D;
    int K;
    const float *logp;   // packed fragment image Lp[ceil(3K/16)][ceil(ldx/16)][64][4]
    const float *cst;    // [3K]: 3k: log w_k ; 3k+1+s: log lr_w[k][s]
    float *scratch;
    int64_t scratch_stride;
    int scratch_by_tile;
    int labels_only;
    int32_t *bins;
    uint64_t seed;
    uint32_t epoch;
    int final_argmax;
    // u8 kernel: previous labels are valid (row-block selection), bin-sorted visiting order of the last statistics pass (nullable)
    int use_prev;
    const int32_t *order;
    const int32_t *order_total;
};
hipError_t launch_mult_sweep(const MultSweepArgs &a, int grid, hipStream_t s);
hipError_t launch_mult_pack(const float *logp, float *Lp, int rows, int64_t ldx, hipStream_t s);
int mult_tile_points();
// bf16 path (count data): exactness check of X, 3-plane bf16 split of the log-probabilities, sweep
hipError_t launch_bf16_exact_check(const float *X, int64_t nwords, int *d_flag, hipStream_t s);
size_t mult_pack_bf16_words(int rows, int64_t ldx);
hipError_t launch_mult_pack_bf16(const float *logp, uint32_t *Lp16, int rows, int64_t ldx, hipStream_t s);
hipError_t launch_mult_sweep_bf16(const MultSweepArgs &a, const uint32_t *Lp16, int grid, hipStream_t s);
// u8 path (integer data in [0, 255]): byte copy of the points, parameter planes in the matching feature order, sweep, statistics
hipError_t launch_u8_convert(const float *X, int64_t ldx, int D, int64_t n, uint8_t *X8, int64_t ld8, int *d_flag, hipStream_t s);
size_t mult_pack_u8_words(int rows, int64_t ld8);
hipError_t launch_mult_pack_u8(const float *logp, uint32_t *Lp8, int rows, int64_t ldx, int64_t ld8, hipStream_t s);
hipError_t launch_mult_sweep_u8(const MultSweepArgs &a, const uint8_t *X8, int64_t ld8, const uint32_t *Lp8, int grid, hipStream_t s);

// ---- label bookkeeping (labels.hip)
// dst/src: device or pinned-host pointers, 4-byte aligned; bytes rounded up to a multiple of 4
hipError_t launch_smart_project(const int32_t *bins, const float *X, int64_t ldx, int64_t n, int D, int k, const double *v, const double *mu,
                                double *proj, double *vals, unsigned long long *counter, hipStream_t s);
hipError_t launch_smart_kmeans(const int32_t *bins, const double *proj, int64_t n, int k, double m_lo, double m_hi, double *partial, double *out,
                               hipStream_t s);
hipError_t launch_smart_assign(int32_t *bins, const double *proj, int64_t n, int k, double m_lo, double m_hi, hipStream_t s);
int smart_groups();
hipError_t launch_predict_finish(const float *table, int64_t stride, int rstep, int64_t n, int K, int64_t *labels, float *probs, hipStream_t s);
hipError_t launch_ingest_rows(float *dst, int64_t ldx, const void *src, int is_f64, int64_t ld, int64_t rows, int D, int nan_to_zero,
                              hipStream_t s);
hipError_t launch_copy_bytes(void *dst, const void *src, size_t bytes, hipStream_t s);
hipError_t launch_copy_bytes16(void *dst, const void *src, size_t bytes, hipStream_t s);   // 16-byte aligned dst / src, size rounded up to 16
hipError_t launch_init_labels(int32_t *bins, int64_t n, int64_t first_index, int init_clusters, int label0, uint64_t seed,
                              uint32_t epoch, hipStream_t s);
// step statistics: flags[k] = 1 when cluster k has an empty sub-cluster (counts: [2K] Int64, global), flags[K] = any;
// then re-draw the sub-labels of flagged clusters (reset_bad_clusters_worker!)
hipError_t launch_bad_flags(const int32_t *bin_total, const long long *global_counts, int K, uint8_t *flags, hipStream_t s);
hipError_t launch_widen_counts(const int32_t *src, int stride, long long *dst, int n, hipStream_t s);
hipError_t launch_gather_rows(float *dst, int64_t ld_dst, const float *src, int64_t ld_src, const int32_t *slot, int rows, int D, hipStream_t s);
hipError_t launch_bins_from_i64(int32_t *bins, const int64_t *labels, const int64_t *sub, int64_t n, hipStream_t s);
hipError_t launch_bins_to_i64(const int32_t *bins, int64_t *labels, int64_t *sub, int64_t n, hipStream_t s);
hipError_t launch_contingency(const int32_t *bins, const int32_t *gt, int64_t n, int K, int n_gt, unsigned long long *counts, hipStream_t s);
hipError_t launch_i64_to_i32(int32_t *dst, const int64_t *src, int64_t n, hipStream_t s);
// pairs: [2*m] = idx[0..m-1], new_idx[0..m-1] as 0-based Int32 cluster ids (device memory)
hipError_t launch_split(int32_t *bins, int64_t n, int64_t first_index, const int32_t *pairs, int m, uint64_t seed,
                        uint32_t epoch, hipStream_t s);
hipError_t launch_merge(int32_t *bins, int64_t n, const int32_t *pairs, int m, hipStream_t s);
hipError_t launch_remap(int32_t *bins, int64_t n, const int32_t *map, hipStream_t s);  // label k -> map[k]
hipError_t launch_reset_sub(int32_t *bins, int64_t n, int64_t first_index, const int32_t *idx, int m, uint64_t seed,
                            uint32_t epoch, hipStream_t s);

// ---- stable counting sort of the points by bin + segmented statistics (suffstats.hip)
constexpr int SORT_TILE = 2048;  // points per sorting wave of big shards (SortBufs::tile: 2048 or SORT_TILE_SMALL)
constexpr int SORT_TILE_SMALL = 512;
constexpr int FAST_TOTAL_STRIDE = 32;   // ints between the running totals of two bins (one 128-byte line per bin: the histogram adds with atomics)

struct SortBufs {
    int tile;             // points per sorting wave of this context's passes: SORT_TILE or SORT_TILE_SMALL
    int32_t *tile_hist;   // [nbins][ntiles_sort] exclusive prefix over the tiles of a bin (written by the scan from tile_cnt)
    int32_t *tile_cnt;    // [nbins][ntiles_sort] points of bin b in tile t (written by the histogram)
    int32_t *spec_bins;   // [n] the re-drawn bin of every point whose cluster was one-sided in the point's tile (written by hist_kernel<.., SPEC>, read by the scatter for flagged clusters)
    int32_t *tile_spec;   // [min(nbins, STEP_SPEC_MAX_BINS)][ntiles_sort] the same counts with the bad-cluster reset of the tile's one-sided clusters counted ahead (hist_kernel<.., SPEC>)
    int32_t *fast_total;  // [nbins][FAST_TOTAL_STRIDE] (element 0 of each line) running bin totals of the per-step histogram (integer atomics), cleared by scan_starts_kernel
    unsigned *ticket;     // [1] unused since the per-step scan and the starts are two launches (kept: the buffers are allocated as a set)
    uint16_t *prev_lab;   // [n] cluster label of every point at the previous per-step pass (0xFFFF: none yet); null: no tracking
    uint8_t *cdirty;      // [DPMM_MAX_CLUSTERS_K + 1] a point entered or left cluster k since then; last element: every cluster
    uint8_t *cmode;       // [DPMM_MAX_CLUSTERS_K] per-step pass: 0 both sub-clusters computed, 1 / 2 left / right derived from the cached cluster row
    int32_t *bin_total;   // [nbins]
    int32_t *bin_start;   // [nbins + 1]
    int32_t *perm;        // [n]
    int32_t *item_start;  // [nbins + 1]
    uint8_t *bin_sel;     // [nbins] 1 = compute statistics for this bin
    int32_t *perm_total;  // [1] number of points placed in perm by the last sort (== n when every label was in range)
};

hipError_t launch_sort_by_bin(const int32_t *bins, int64_t n, int nbins, const SortBufs &b, hipStream_t s);
constexpr int STEP_SPEC_MAX_BINS = 512;      // bins (2K) up to which the per-step histogram counts the bad-cluster reset ahead (a second set of LDS counters per tile)
struct StepReset { const long long *global_counts; uint8_t *flags; int K; uint8_t *cside; int64_t first; uint64_t seed; uint32_t epoch; };
hipError_t launch_step_hist(const int32_t *bins, int64_t n, int nbins, const SortBufs &b, hipStream_t s, int spec = 0, int64_t first = 0, uint64_t seed = 0, uint32_t epoch = 0);
hipError_t launch_step_reset(int32_t *bins, int64_t n, int64_t first, int nbins, const SortBufs &b, const long long *global_counts, uint8_t *flags,
                             int K, uint64_t seed, uint32_t epoch, uint8_t *cside, hipStream_t s);     // cside != null: speculative reset of this shard's candidates
// rows [2K][stride] of the speculatively reset labels -> red [3K][stride] (the travelling rows of the one-collective pass; Multinomial: its reduce does not write them itself)
hipError_t launch_onecoll_rows(const double *rows, double *red, int64_t stride, int K, const uint8_t *cside, uint8_t *flags, hipStream_t s);
hipError_t launch_niw_finalize_rows(const double *red, double *out, int64_t stride, int K, const uint8_t *cside, uint8_t *flags, uint8_t *flags_host, hipStream_t s);
hipError_t launch_niw_undo_reset(int32_t *bins, int64_t n, int K, const uint8_t *flags, const uint8_t *cside, hipStream_t s);
struct StatsArgs;
hipError_t launch_sort_finish(const int32_t *bins, const StatsArgs &a, hipStream_t s);
hipError_t launch_step_scan_scatter(int32_t *bins, const StatsArgs &a, int derive, int force_all, int fused_starts, const StepReset *rs, hipStream_t s);
hipError_t launch_derive_rows(double *out, double *cache, const uint8_t *mode, uint8_t *dirty, int64_t stride, int K, const uint8_t *flags_src,
                              uint8_t *flags_dst, hipStream_t s);

struct StatsArgs {
    const float *X;
    int64_t ldx;
    int64_t n;
    int D;
    int nbins;
    int chunk;             // points per work item
    int max_items;
    int range_groups;      // NIW: workgroups of the statistics kernel; each owns a contiguous range of items (0: one item per workgroup)
    SortBufs sb;
    double *slabs;         // NIW: [NIW_STATS_MAX_GROUPS + nbins][slab_stride] (slots, suffstats.hip head_slot); Multinomial: [max_items][slab_stride]
    int64_t slab_stride;
    double *out;           // packed [nbins][packed_stride]
    int64_t packed_stride;
    const int32_t *row_off; // NIW: [packed_stride] packed-row element -> position inside a slab (launch_niw_row_offsets)
    const int32_t *inv_off; // NIW: [slab_stride] slab position -> packed-row element (-1: none), the inverse table
    // NIW per-step pass with derived statistics: the reduce kernel forms the rows that were not computed (null mode: plain reduce)
    const uint8_t *mode;    // [K] 0 both computed -> cache = left + right, 1 / 2: left / right = cache - computed
    double *cache;          // [K][packed_stride]
    uint8_t *dirty;         // [DPMM_MAX_CLUSTERS_K + 1] cleared for the next pass
    const uint8_t *flags_src; uint8_t *flags_dst; int K;    // rider: bad-cluster flags -> the caller's pinned block (null: none)
    // one-collective per-step pass: `out` has 3K rows -- [2K rows of the labels as swept | K re-drawn left rows] -- cside [K] says which
    // clusters' sub-labels were reset speculatively on this shard (suffstats.hip reset_recount_kernel); zero2 = two flag bytes to clear
    const uint8_t *cside; uint8_t *zero2;
};
constexpr int NIW_STATS_MAX_GROUPS = 4096;   // workgroups of the NIW statistics kernel at most; slab slots = this + 2 K (suffstats.hip head_slot)
int64_t niw_slab_stride(int D);
hipError_t launch_niw_row_offsets(int32_t *row_off, int32_t *inv_off, int D, int64_t packed_stride, hipStream_t s);
int64_t mult_slab_stride(int D);
hipError_t launch_niw_stats(const StatsArgs &a, hipStream_t s);
hipError_t launch_mult_stats(const StatsArgs &a, hipStream_t s);
hipError_t launch_mult_stats_u8(const StatsArgs &a, const uint8_t *X8, int64_t ld8, hipStream_t s);

// ---- the Multinomial master's draws on the device (mult_master.hip) ----
#define DPMM_MULT_MASTER_MAXD 16384      // the row's log Gamma variates live in LDS (8 bytes each)
hipError_t launch_mult_dirichlet(const double *rows, int64_t stride, const float *alpha0, const float *alpha1, int outlier_first, int D, int64_t ldx,
                                 int K, uint64_t seed, uint32_t epoch, float *raw, hipStream_t s);
#define DPMM_MULT_MASTER_MAXPAIRS 8192
hipError_t launch_mult_marginals(const double *rows, int64_t stride, const float *alpha0, const float *alpha1, int outlier_first, int D, int K,
                                 const int32_t *pairs, int npairs, const double prior_c[4], double *out, hipStream_t s);

// ---- the master's dense maths on the device (niw_master.hip) ----
#define DPMM_MASTER_MAXD 256
#ifndef DPMM_MASTER_NSCALARS
#define DPMM_MASTER_NSCALARS 8      // doubles per distribution in the scalar records: N, kappa', nu', log det(nu' psi'), log Gamma_D(nu' / 2), 3 spare (dpmm_hip.h)
#endif
struct NiwMasterArgs {
    int D, DP;                    // DP = 16 * ceil(D / 16)
    int64_t packed_stride;
    double kappa0, nu0;
    const double *m0;             // [D]
    const double *psi_lo;         // prior psi, symmetrised, packed lower triangle [D (D + 1) / 2]
    double *fac;                  // [rows][DP][DP]  P = nu' psi' -> its factor L (row = 3 slot + w)
    double *mean;                 // [rows][DP]      m'
    double *kap, *nu;             // [rows]
    double *rows_store;           // [slots][2][packed_stride]  statistics of every slot (left, right)
    float *mu_draw;               // [3 K][DP]       the current draws' means (cluster order)
    uint64_t seed;
};
size_t niw_master_lds_bytes(int DP);
hipError_t launch_niw_master_pairs(const NiwMasterArgs &a, const int32_t *pairs, int n, double *scratch, double *small, hipStream_t s);
hipError_t launch_niw_rows_gather(const double *rows_store, const int32_t *slots, int n, int64_t stride, double *dst, hipStream_t s);
hipError_t launch_niw_master_posterior(const NiwMasterArgs &a, const int32_t *jobs, int njobs, const double *rows, double *small, hipStream_t s);
bool niw_master_can_fuse_pairs(const NiwMasterArgs &a);
hipError_t launch_niw_master_posterior_pairs(const NiwMasterArgs &a, const int32_t *jobs, int njobs, const double *rows, double *small,
                                             const int32_t *cluster_pairs, int npairs, double *pair_small, hipStream_t s,
                                             int noise_nmat = 0, uint32_t noise_epoch = 0, double *noise_Y = nullptr);   // noise_Y: also the standard normals of the draws that follow (round 6)
hipError_t launch_niw_draw_inputs(const NiwMasterArgs &a, const int32_t *slot_of_cluster, int K, uint32_t epoch, double *Aout, double *xiout, hipStream_t s);
hipError_t launch_niw_master_noise(const NiwMasterArgs &a, int nmat, uint32_t epoch, double *Y, hipStream_t s);
hipError_t launch_niw_master_draw(const NiwMasterArgs &a, const int32_t *slot_of_cluster, int K, uint32_t epoch, double *Y, float *logdet_sigma,
                                  const float *lr, const float *wts, float *Rp, float *mup, float *cst, float *tail, int NB,
                                  unsigned long long *work, int what, hipStream_t s);

// Test hook (dpmm_debug_set_prelaunch_hook): called on the host in front of EVERY kernel launch of the library.  tests/tools/poison.py
// installs a function that waits for the device and refills the LDS and the register files of every CU with a NaN pattern, so that a
// kernel reading LDS or registers it never wrote sees that instead of the (finite) leftovers of the library's own previous kernel.
extern void (*g_prelaunch)(void *);
extern void *g_prelaunch_arg;
inline void prelaunch() { if (g_prelaunch) g_prelaunch(g_prelaunch_arg); }
#define DPMM_LAUNCH(...) do { ::dpmm::prelaunch(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

}  // namespace dpmm
