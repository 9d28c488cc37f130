// dpmm_api.cpp -- C ABI of libdpmmhip.so (see include/dpmm_hip.h -- and its companions dpmm_hip_master.h, dpmm_hip_debug.h -- for the contract and the
// reference functions each entry point replaces).  Host-side glue only: device memory
// ownership, parameter staging, kernel sequencing on ONE stream, error mapping.
// There is no CPU fallback anywhere in this file: without a usable gfx950 device
// dpmm_create fails with DPMM_ENODEVICE.
#include "../../include/dpmm_hip.h"
#include "../../include/dpmm_hip_master.h"
#include "../../include/dpmm_hip_debug.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <string>
#include <unordered_map>
#include <vector>

#include "dpmm_kernels.h"

namespace dpmm {      // test hook in front of every kernel launch (dpmm_kernels.h DPMM_LAUNCH, dpmm_debug_set_prelaunch_hook)
void (*g_prelaunch)(void *) = nullptr;
void *g_prelaunch_arg = nullptr;
}
using namespace dpmm;

#ifdef DPMM_POISON
// Diagnostic build (scripts/build_variant.sh poison -DDPMM_POISON=0xFF): every device allocation of this file is filled with the poison byte
// before anybody uses it -- a kernel that reads memory nobody wrote then reads NaNs / -1 on EVERY box, not only on one whose memory holds
// another process's leftovers.  (Built while a one-in-twenty chain divergence on fresh boxes was tracked down: it ruled device MEMORY out;
// the cause was a never-written LDS word, found with tests/tools/poison.py -- DESIGN section 5.)
template <typename T>
static hipError_t hipMallocPoisoned(T **p, size_t n) {
    hipError_t e = hipMalloc(p, n);
    if (e == hipSuccess && n > 0) { e = hipMemset(*p, DPMM_POISON, n); if (e == hipSuccess) e = hipDeviceSynchronize(); }
    return e;
}
#define hipMalloc hipMallocPoisoned
#endif

static_assert(DPMM_MAX_CLUSTERS == DPMM_MAX_CLUSTERS_K, "header / kernel limits out of sync");

#ifdef DPMM_STAMPS
static unsigned long long *g_dbg = nullptr;
#endif
namespace {
thread_local std::string g_create_error;

int nb_for_dim(int D) { return D <= 16 ? 1 : D <= 32 ? 2 : D <= 64 ? 4 : D <= 128 ? 8 : 16; }
}  // namespace

struct dpmm_ctx {
    int prior = 0, D = 0, device = 0;
    int64_t n = 0, first = 0, ldx = 0;
    uint64_t seed = 0;
    int NB = 0;          // NIW: 16-blocks per dimension
    int tile = 0;        // points per sweep workgroup tile
    int64_t ntiles = 0;
    int cus = 256;
    int sweep_grid = 0, sweep_grid_max = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_part[2] = {nullptr, nullptr};   // DPMM_OPT_KERNEL_TIMING bit 3: behind niw_lean_kernel / behind the LSTORE sweep kernel (dpmm_last_sweep_parts_ms)
    int have_parts = 0;                           // 0: no three-kernel sweep timed yet; 1: labels + sub-labels (no lean kernel); 2: lean + labels + sub-labels
    bool have_sweep_ev = false, have_stats_ev = false;

    float *dX = nullptr;
    int32_t *dbins = nullptr;
    bool have_points = false, have_labels = false;
    int32_t *d_gt = nullptr;  // ground truth of the shard (on-device evaluation)
    int n_gt = 0;
    unsigned long long *d_cont = nullptr;
    bool have_perm = false;   // sb.perm holds a permutation of [0,n) sorted by a recent labelling

    // parameters
    int K = 0, Kcap = 0;
    float *d_raw = nullptr;   // NIW: R [3K][D][D] ; MULT: unused
    float *d_mu = nullptr;    // NIW raw mu [3K][D]
    float *d_Rp = nullptr;    // NIW packed fragments / MULT packed logp
    float *d_mup = nullptr;   // NIW padded mu
    float *d_cst = nullptr;   // [3K]
    uint32_t *d_Lp16 = nullptr;  // MULT: 3-plane bf16 split of the log-probabilities (count data fast path)
    // MULT, device master: the NEXT parameter draws, made right behind the per-step statistics (while the host decides splits and merges) into
    // a second set of buffers; dpmm_mult_master_draw swaps the sets when its epoch / K / outlier flag are the ones guessed and the rows have
    // not changed since (else it draws as before)
    float *d_raw2 = nullptr, *d_Rp2 = nullptr;
    uint32_t *d_Lp16_2 = nullptr;
    bool mspec_valid = false;
    bool rp_current = true, rp2_current = true;   // the Float32 fragment image (d_Rp / d_Rp2) matches d_raw / d_raw2: the device master skips it for byte / bf16 data; the Float32 sweep packs it on demand
    uint32_t mspec_epoch = 0, mdraw_epoch = 0;
    int mspec_K = 0, mspec_outlier = 0, opt_mult_draws_ahead = 1;
    bool mdraw_seen = false;           // a dpmm_mult_master_draw has run: its epoch + 1 is the guess
    long long mspec_used = 0;          // draws taken from the set made ahead (dpmm_debug / tests)
    hipEvent_t ev_rows = nullptr;      // dpmm_step_stats: rows and flags are in the pinned block (the draws launched ahead follow it)
    bool rows_on_demand = false, rows_late = false;   // dpmm_mult_master_rows_on_demand: with draws launched ahead the rows reach the pinned block BEHIND them
    // dpmm_mult_master_draw re-uses its pinned weights once the host has waited for the main stream (or for ev_rows, recorded later on it) since
    // the copy that read them -- a generation count instead of an event of its own (an event record costs ~5 us of stream time in front of the sweep)
    unsigned long long sync_gen = 0, cst_gen = 0;
    bool cst_inflight = false;
    bool marg_behind_ev = false;       // the log-marginals in the pinned block were complete at ev_rows (no need to wait for the stream)
    int x_bf16_exact = 0;     // MULT: every x is exactly representable in bf16 (checked at upload)
    uint8_t *dX8 = nullptr;   // MULT: byte copy of the points when every x is an integer in [0, 255] ([n][ld8])
    int64_t ld8 = 0;
    int x_u8 = 0;
    int opt_no_u8 = 0;
    float *d_scratch = nullptr;
    int64_t scratch_stride = 0;
    bool have_params = false;
    float *d_lam = nullptr, *d_mdist = nullptr, *d_tail = nullptr;   // NIW screening constants (per sweep)
    bool have_screen_prep = false, have_tail = false;
    float *d_tdf = nullptr;   // Student-t constants of the predictive mode ([3K][2]) or null
    bool predictive = false;

    // sort + stats
    SortBufs sb{};
    int nt_sort = 0;          // tiles the sort tables are allocated for
    int sort_tile_min = SORT_TILE;   // smallest sort tile the tables can hold
    int chunk = 512;
    int max_items = 0;
    double *d_slabs = nullptr;
    int64_t slab_stride = 0;
    int32_t *d_row_off = nullptr;      // NIW: packed-row element -> slab position
    int32_t *d_inv_off = nullptr;      // NIW: slab position -> packed-row element (reduce kernel)
    double *d_out = nullptr;
    int64_t packed_stride = 0;
    double *d_proj = nullptr, *d_vals = nullptr, *d_smart = nullptr;   // smart splits: projections [n], compacted copy [n], partials + v + mu
    int32_t *d_small = nullptr;  // index lists for relabel kernels (Int32, <= 4*DPMM_MAX_CLUSTERS)
    std::vector<uint8_t> h_sel;
    // pinned host staging for every per-step transfer (pageable copies stall for tens of ms now and then)
    char *h_pin = nullptr;
    size_t h_pin_bytes = 0;
    // parameter staging (dpmm_params_staging): slot-indexed rows the host writes in place; the pack kernels read it directly
    char *h_par = nullptr;
    size_t h_par_bytes = 0;
    int par_slots = 0;
    // statistics output (dpmm_step_stats / dpmm_suffstats_host): packed rows + flags, read by the host in place
    char *h_out = nullptr;
    size_t h_out_bytes = 0;
    char *d_par = nullptr;             // device copy of the staging's mu | R regions (NIW): the pack kernel gathers from HBM, not over the host link
    size_t d_par_bytes = 0;
    bool work_zeroed = false;          // the pack kernel cleared d_work and no sweep has run since
    int sel_all_ones = 0, sel_capacity = 0;   // sb.bin_sel[0..sel_all_ones) are known to be 1 (full passes skip the memset)
    long long *d_counts64 = nullptr;   // [2 * DPMM_MAX_CLUSTERS] global sub-cluster occupancies (multi-GPU)
    // Multinomial master on the device (mult_master.hip): priors, and how many clusters' complete rows the last statistics pass left in d_out
    float *d_malpha = nullptr;         // [2][ldx]: cluster prior | outlier prior
    bool mult_master = false, mult_has_alpha1 = false;
    int rows_full_K = -1;
    // ... and its log-marginals (mult_marginal_kernel): prior constants {sum a, sum lgamma(a)} x 2, the pair list asked for ahead of the
    // next per-step pass, the pinned result block [3K][2] (N, L) | [pairs] and what it currently answers
    double mult_prior_c[4] = {0, 0, 0, 0};
    int32_t *d_mpairs = nullptr;
    std::vector<int32_t> mpairs_req, mpairs_shadow;
    bool marg_req = false, marg_valid = false;
    int marg_req_outlier = 0, marg_K = 0, marg_np = 0;
    double *h_marg = nullptr;
    // derived sub-cluster statistics of the per-step pass (derive_rows_kernel): cached cluster-level rows + label tracking
    double *d_ccache = nullptr;        // [Kcap][packed_stride] left + right of every cluster as of the last pass that computed both
    bool cache_force = true;           // the next per-step pass computes every cluster in full (points uploaded, cache re-allocated, K changed)
    int cache_K = -1;
    int opt_derive = 1;
    int64_t dbg_early_wait = 0;        // event waits that returned before the posteriors' records were in host memory (dpmm_debug_counters)
    int opt_noise_ahead = -1;          // normals of the next draws on the second stream beside the sweep (DPMM_OPT_NOISE_AHEAD): -1 = for D >= 128 or shards below 4e6 points
    // device master (niw_master.hip)
    bool master = false;
    NiwMasterArgs ma{};
    double *d_m0 = nullptr, *d_psi_lo = nullptr, *d_pairs = nullptr;
    double *d_Y[2] = {nullptr, nullptr};           // draw outputs, two sets: [draw_cur] belongs to the parameters in use, the other takes the next draws
    float *d_mu_draw[2] = {nullptr, nullptr};
    int draw_cur = 0;
    size_t pair_cap = 0;                           // matrices in d_pairs (pooled pair scratch)
    float *d_ld_sigma[2] = {nullptr, nullptr};
    int master_slots = 0, master_K = 0;            // capacities: slots (fac / mean / rows_store), clusters (Y / mu_draw)
    // draws launched ahead (dpmm_step_master_device): a second stream, so that the pair kernels of the merge proposals do not queue behind them
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_master = nullptr, ev_spec = nullptr;      // posteriors done (main stream) / early draws done (stream2)
    hipEvent_t ev_noise = nullptr;                          // normals of the next draws generated (stream2)
    // pooled pair log-determinants launched ahead (dpmm_step_master_device): list in device memory, records in a pinned block of their own
    hipEvent_t ev_pairs = nullptr;
    int32_t *d_apairs = nullptr;                            // [2 cap] slot pairs of the launch-ahead job
    size_t apairs_cap = 0;
    std::vector<int32_t> apairs_shadow;
    int32_t *h_apairs_list[2] = {nullptr, nullptr}; size_t h_apairs_list_cap[2] = {0, 0}; int apairs_pin_flip = 0; const int32_t *apairs_pinned_cur = nullptr;      // the fused pair jobs' list in pinned memory (DPMM_OPT_CHAIN_FUSION bit 4)
    double *h_apairs = nullptr;                             // pinned [cap][DPMM_MASTER_NSCALARS]
    bool apairs_inflight = false, apairs_valid = false;     // main stream has not waited for ev_pairs yet / the records answer dpmm_niw_master_pairs
    std::unordered_map<uint32_t, int> apairs_index;         // (slot_i << 16 | slot_j) -> record
    std::vector<uint8_t> apairs_dirty;                      // [slot] statistics of the slot changed since the job was launched
    std::vector<int32_t> apairs_req;                        // request of dpmm_niw_master_pairs_ahead, consumed by the next dpmm_step_master_device
    char *h_draw = nullptr;                                 // pinned: lr [K][2] | w [K] of dpmm_niw_master_draw
    bool handover_inflight = false;                         // a hand-over kernel (reads lr / w from h_pin) was launched and the host has not waited behind it
    bool noise_pending = false;                             // dpmm_niw_master_draw asked for the next normals; launched behind the sweep (noise_flush)
    uint32_t noise_pend_epoch = 0;
    int noise_pend_nmat = 0;
    bool noise_inflight = false, noise_valid = false;       // main stream has not waited for ev_noise yet / d_Y[noise_buf] holds the normals of noise_epoch
    uint32_t noise_epoch = 0;
    int noise_nmat = 0, noise_buf = 0;
    bool spec_inflight = false, spec_valid = false;         // main stream has not waited for ev_spec yet / the early draws are still the ones a draw call would make
    uint32_t spec_epoch = 0;
    std::vector<int32_t> spec_slots;
    // index lists of the master kernels (jobs, slot maps) live in device memory and are re-sent only when they change: read from pinned
    // host memory they cost every workgroup a PCIe round trip (~2 us) before its first useful instruction
    int32_t *d_jobs = nullptr, *d_dslots = nullptr;          // [2 MAX] jobs of the posterior kernels (main stream) / [MAX] slot map of the draws
    int32_t *h_list[4] = {nullptr, nullptr, nullptr, nullptr};   // pinned staging ring of device_list
    size_t h_list_cap[4] = {0, 0, 0, 0};
    int h_list_next = 0;
    std::vector<int32_t> jobs_shadow, dslots_shadow;
    uint8_t *h_master = nullptr;                   // pinned: jobs | slot map | lr | w | small
    size_t h_master_bytes = 0;
    bool draws_on_device = false;
    unsigned long long *d_work = nullptr;   // [DPMM_WORK_SLOTS + 4 DPMM_WORK_PER_WAVE sweep_grid_max]: tile queue heads [4], [8 + 16 q]; [DPMM_WORK_SLOTS + DPMM_WORK_PER_WAVE w ..] executed-work counters of wave w of the last sweep
    int work_waves = 0;                      // most waves of a counted sweep launch since the counters were read
    long long work_launches = 0;             // counted sweep launches since the counters were read
    // options (dpmm_set_option)
    float opt_margin = 50.f;
    int opt_prio = 1;
    int opt_queue_rounds = -1;
    int opt_ball = 1;
    // direction screen of the D in 33..64 sweep (direction_far, niw_sweep.hip; DPMM_OPT_DIRECTION_SCREEN): tables per parameter set, built only while
    // the sweeps report tiles with many candidates (h_need: one word per wave of the last sweep, written by the kernel into pinned memory)
    uint32_t *d_sp_frag = nullptr;
    uint32_t *d_refb_big = nullptr;    // D = 128, 256: bf16 images of the cluster-level factors (reference bracket of the LDS-staged kernels), per parameter set
    bool have_refb_big = false;
    uint32_t *d_brk_flag = nullptr;    // [tiles of 128 points] 1 + k0 where the tile is label-homogeneous, else 0 (niw_bracket_big_kernel)
    float *d_brk_aref = nullptr;       // [positions of the visiting order] the bracket's lower end of a_k0
    float *d_sp_cons = nullptr;
    uint32_t *h_need = nullptr;
    int opt_direction = -1;             // -1: by the previous sweep's candidate counts, 0: never, 1: always
    bool sp_ready = false, sp_regime = false;
    unsigned sp_count = 0;             // parameter sets since the screen came on: every 32nd sweep measures (tail pairs first), the others run it first
    bool sp_last = false;              // the last sweep ran the DIR kernel (its yield words are valid)
    int sp_K = -1;                     // number of clusters of the last parameter set the tables were built for
    int sp_cooldown = 0;               // parameter sets the screen stays off after it removed less than a quarter of what it was given
    int opt_b3 = 1;                    // DPMM_OPT_B3_SUBLABELS: D in 33..64: sub-cluster evaluations through three-plane bf16 images, in kernels of their own (niw_lean.hip)
    bool have_b3 = false;              // the images behind the bracket's in d_tail belong to the parameter set on the device
    int opt_chain = 0x7fffffff & ~(8 | 32);  // (bits 8 / 32: built, value-neutral, measured -- no gain at either size: off by default) DPMM_OPT_CHAIN_FUSION (bit mask): 1 = the sort's starts inside the scatter launch; 2 = the three-plane images in the hand-over launch; 4 = the fused pair jobs' list read from pinned memory; 8 = the bad-cluster reset counted ahead by the histogram and applied by the scatter (no reset launch); 16 = the draws' normals generated inside the posteriors' launch
    int opt_pair_ball = 1;             // DPMM_OPT_PAIR_BALL: the lean kernel's full-dimension ball test on the K x K table of pair distances (niw_pair_ball_kernel, K <= 256)
    bool have_pb = false;              // ... and that table belongs to the parameter set on the device
    int opt_lean_dir = 1;              // DPMM_OPT_LEAN_DIRECTION: the lean kernel runs the direction screen while its tables exist (0: no lean launch in that regime, as in rounds 4-5)
    int opt_master_poll = 1;           // DPMM_OPT_MASTER_POLL: dpmm_step_master_device waits on the posteriors' own records in pinned memory (no event between posteriors and draws)
    int opt_lean = 1;                  // DPMM_OPT_LEAN_TILES: tiles the cheap screens settle completely in niw_lean_kernel (-1 automatic is 1 with a regime switch; 0 never)
    uint32_t *d_hard = nullptr;        // two lists of [2 + ceil(n / 64)] words, taking turns (hard_flip): count | wave tiles the lean kernel left to the general path;
                                       // a lean launch clears the OTHER list's count for its successor (no fill launch), niw_sub_kernel reports the count to h_hard (no copy launch)
    int hard_flip = 0;
    int perm_nbins = 0;                // bins of the sort that wrote sb.perm / sb.bin_start (0: none yet): the lean kernel aligns its tiles to them
    uint32_t *h_hard = nullptr;        // pinned: the count of the LAST sweep's list (read by the next sweep's regime decision, never waited for)
    int lean_off = 0;                  // sweeps left without the lean kernel (a sweep that left more than 30 % of its tiles switches it off for lean_backoff)
    bool lean_ran = false;             // the last sweep ran the lean kernel: h_hard holds its list's length
    int lean_backoff = 15;             // 15, doubled by every retry that fails again (up to 1023), back to 15 by one that succeeds: a retry on overlapping
                                       // clusters costs three sweeps' time (MixtureVar 4: 5.3 ms against 1.8), one in 16 was 0.22 ms per step
    int opt_bf16scr = 1;               // D <= 64 sweep: bf16 screens in front of the Float32 16-row screen / of a survivor's first row block (DPMM_OPT_BF16_SCREENS)
    int opt_bracket = 1;               // D <= 64 sweep: certified bf16 bracket of the reference cluster's value instead of its Float32 evaluation where that decides nothing (DPMM_OPT_REF_BRACKET)
    int opt_timing = 0;                // bit 0 / 1 / 2: HIP events around the sweep kernel / the statistics pass / the all-reduces (dpmm_last_kernel_ms, dpmm_last_comm_ms)
    int opt_tail = 1, opt_prescreen = -1, opt_ordered = 1, opt_force_f32 = 0, opt_trace = 0, opt_ref_const = 0;
    int64_t opt_stats_items = 0;
    int opt_stats_groups = 0;
    // collective (dpmm_comm_init: RCCL on the ctx stream; dpmm_comm_init_host: a caller-supplied host all-reduce)
    void *comm = nullptr;
    int rank = 0, world = 1;
    dpmm_host_allreduce_fn host_fn = nullptr;
    void *host_user = nullptr;
    char *h_red = nullptr;             // pinned staging of the host transport
    size_t h_red_bytes = 0;
    hipEvent_t ev_comm[4] = {nullptr, nullptr, nullptr, nullptr};   // [0,1] around the occupancy all-reduce, [2,3] around the packed rows
    bool have_comm_ev[2] = {false, false};
    int64_t comm_bytes[2] = {0, 0};    // payload of the last all-reduce of each kind
    int64_t comm_calls = 0;            // all-reduces since the communicator was attached
    // Bounded waits behind a collective (RCCL transport; DPMM_OPT_COMM_TIMEOUT_MS): a rank that dies between two all-reduces leaves the
    // others inside an RCCL kernel that never ends -- and the reference has the same flaw (its master waits on `fetch`).
    // A watchdog thread per communicator measures how long the host has been blocked on the ctx stream BEHIND AN ENQUEUED COLLECTIVE
    // (`coll_pending`: set when an all-reduce is enqueued, cleared when a wait for the whole stream has returned -- waits on a stream that
    // holds no collective, e.g. an upload or a checkpoint, are never timed); past the deadline it aborts the communicator (ncclCommAbort:
    // the kernel returns), the blocked call comes back and fails with DPMM_ECOMM -- on every surviving rank -- instead of hanging until
    // somebody kills the job.  A peer that is merely SLOW (a long host-side pause between two steps) looks the same from here as one that
    // is gone, and the reference would simply wait: the default limit is therefore half an hour, far beyond any pause of a healthy job.
    // `wd_wait_since` is written, and the abort performed, under `wd_mu`: the host thread that comes back from its wait either clears the
    // mark before the watchdog looks, or finds the communicator aborted -- it never enqueues on a communicator that is being aborted.
    // After an abort only the collectives refuse (DPMM_ECOMM); entry points that use none (dpmm_get_labels, dpmm_sync) keep working, so
    // that the state can still be saved.  The host transport's callback owns its own time-out.
    int comm_timeout_ms = 1800000;
    bool coll_pending = false;         // an RCCL collective was enqueued and no full wait on the ctx stream has returned since
    bool abort_reported = false;       // the call that was blocked when the watchdog fired has returned DPMM_ECOMM
    std::thread *watchdog = nullptr;
    std::mutex wd_mu;
    std::condition_variable wd_cv;
    bool wd_stop = false;
    std::atomic<int64_t> wd_wait_since{0};      // steady-clock ms when the host began to block on the stream; 0: not blocked
    std::atomic<bool> comm_aborted{false};
    // ONE collective per per-step pass (DPMM_OPT_ONE_COLLECTIVE; NIW, communicator attached): see run_stats
    int opt_one_collective = -1;       // -1: automatic (rows short enough that half a message more is cheaper than a second collective), 0 / 1
    uint8_t *d_cside = nullptr;        // [DPMM_MAX_CLUSTERS] clusters whose sub-labels this shard reset speculatively: the side its points were on (1 / 2), 0: none
    double *d_red = nullptr;           // [3 Kcap][packed_stride]: what travels -- 2K rows of the labels as swept | K re-drawn left rows -- input of the finalize kernel
    bool last_pass_one_collective = false;
    bool undo_pending = false;         // niw_undo_reset_kernel of the last one-collective pass not launched yet (flush_undo)

    std::string err;
};

static inline double now_ms() {
    timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
#define HIPCHK(ctx, expr)                                                                               \
    do {                                                                                                \
        const double t0__ = (ctx)->opt_trace ? now_ms() : 0.0;                                          \
        hipError_t e__ = (expr);                                                                        \
        if ((ctx)->opt_trace) {                                                                           \
            const double dt__ = now_ms() - t0__;                                                        \
            if (dt__ > 5.0) fprintf(stderr, "[dpmm slow] %.2f ms in %s (line %d)\n", dt__, #expr, __LINE__); \
        }                                                                                               \
        if (e__ == hipErrorLaunchTimeOut && (ctx)->comm_aborted.load(std::memory_order_relaxed)) {      /* (sync_stream / sync_event: the watchdog fired during this wait) */ \
            (ctx)->err = "collective timed out: a peer rank is gone or stuck (communicator aborted after DPMM_OPT_COMM_TIMEOUT_MS)"; \
            return DPMM_ECOMM;                                                                          \
        }                                                                                               \
        if (e__ != hipSuccess) {                                                                        \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                            \
            return DPMM_EHIP;                                                                           \
        }                                                                                               \
    } while (0)

static inline int64_t steady_ms() {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// Blocking waits on the ctx stream go through these two: with an RCCL communicator attached the watchdog sees how long they last.
static inline void wd_arm(dpmm_ctx *c) {
    std::lock_guard<std::mutex> lk(c->wd_mu);
    c->wd_wait_since.store(steady_ms() | 1, std::memory_order_relaxed);
}
// true: the watchdog aborted the communicator while this wait was armed (reported once, by the call that was blocked)
static inline bool wd_disarm(dpmm_ctx *c) {
    std::lock_guard<std::mutex> lk(c->wd_mu);      // (an abort in progress finishes first: the caller then sees it)
    c->wd_wait_since.store(0, std::memory_order_relaxed);
    if (c->comm_aborted.load() && !c->abort_reported) { c->abort_reported = true; return true; }
    return false;
}
static inline hipError_t sync_stream(dpmm_ctx *c, hipStream_t st) {
    if (st == c->stream) c->sync_gen += 1;        // (everything queued on the main stream so far has run when this returns)
    if (!c->watchdog || !c->coll_pending) return hipStreamSynchronize(st);
    wd_arm(c);
    hipError_t e = hipStreamSynchronize(st);
    if (wd_disarm(c)) e = hipErrorLaunchTimeOut;
    else if (e == hipSuccess && st == c->stream) c->coll_pending = false;      // the stream is empty: no collective left on it
    return e;
}
static inline hipError_t sync_event(dpmm_ctx *c, hipEvent_t ev) {
    if (!c->watchdog || !c->coll_pending) return hipEventSynchronize(ev);
    wd_arm(c);
    hipError_t e = hipEventSynchronize(ev);
    if (wd_disarm(c)) e = hipErrorLaunchTimeOut;
    return e;
}

static int ensure_pinned(dpmm_ctx *c, size_t bytes) {
    if (bytes <= c->h_pin_bytes) return DPMM_OK;
    HIPCHK(c, sync_stream(c, c->stream));
    if (c->h_pin) hipHostFree(c->h_pin);
    c->h_pin = nullptr; c->h_pin_bytes = 0;
    size_t cap = 1 << 20;
    while (cap < bytes) cap *= 2;
    HIPCHK(c, hipHostMalloc((void **)&c->h_pin, cap, hipHostMallocDefault));
    c->h_pin_bytes = cap;
    return DPMM_OK;
}

static int fail(dpmm_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------------
// The library must load (and `pytest -m "not gpu"` must be able to check its exports) on machines without RCCL, and a
// process that also uses torch.distributed must end up with ONE copy of librccl: the symbols are looked up with dlopen,
// first among the libraries already mapped, then by soname, then under /opt/rocm.
struct UidByValue { char internal[128]; };
namespace {
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value */ UidByValue, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
}  // namespace
static std::string g_rccl_path;   // dpmm_comm_use_library
static Rccl &rccl() {
    static Rccl r;
    if (r.handle || !r.err.empty()) return r;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (!g_rccl_path.empty()) r.handle = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) for (const char *n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.handle) break; }
    if (!r.handle) for (const char *n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.handle) break; }
    if (!r.handle) { r.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return r; }
    auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.err.empty()) r.err = std::string("missing RCCL symbol ") + n; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.handle, "ncclCommAbort"));      // (optional: the watchdog needs it)
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
}
enum { kNcclInt64 = 4, kNcclFloat64 = 8, kNcclInt8 = 0, kNcclSum = 0 };   // ncclDataType_t / ncclRedOp_t values of rccl.h

static inline bool comm_attached(const dpmm_ctx *c) { return c->comm != nullptr || c->host_fn != nullptr; }
// In-place all-reduce(sum) of a device buffer over the ranks, stream-ordered on the ctx stream.  `kind`: 0 = Int64 sub-cluster
// occupancies, 1 = Float64 packed rows (HIP events around each kind: dpmm_last_comm_ms).  Transport: RCCL (dpmm_comm_init) or the
// caller's host function (dpmm_comm_init_host: device -> pinned, synchronise, fn, pinned -> device).
static int comm_allreduce(dpmm_ctx *c, void *dbuf, size_t count, int kind) {
    const bool f64 = kind == 1;
    if (c->opt_timing & 4) {
        if (!c->ev_comm[0]) for (auto &e : c->ev_comm) HIPCHK(c, hipEventCreate(&e));
        HIPCHK(c, hipEventRecord(c->ev_comm[2 * kind], c->stream));
    }
    if (c->comm) {
        Rccl &r = rccl();
        if (c->comm_aborted.load()) return fail(c, DPMM_ECOMM, "the communicator was aborted after a collective timed out: attach a new one (dpmm_comm_release, dpmm_comm_init)");
        const int rc = r.AllReduce(dbuf, dbuf, count, f64 ? kNcclFloat64 : kNcclInt64, kNcclSum, c->comm, c->stream);
        if (rc != 0) return fail(c, DPMM_ECOMM, std::string("ncclAllReduce: ") + r.GetErrorString(rc));
        c->coll_pending = true;
    } else {
        const size_t bytes = count * 8;
        if (bytes > c->h_red_bytes) {
            HIPCHK(c, sync_stream(c, c->stream));
            if (c->h_red) hipHostFree(c->h_red);
            c->h_red = nullptr; c->h_red_bytes = 0;
            size_t cap = 1 << 16;
            while (cap < bytes) cap *= 2;
            HIPCHK(c, hipHostMalloc((void **)&c->h_red, cap, hipHostMallocDefault));
            c->h_red_bytes = cap;
        }
        HIPCHK(c, launch_copy_bytes(c->h_red, dbuf, bytes, c->stream));
        HIPCHK(c, sync_stream(c, c->stream));
        const int rc = c->host_fn(c->host_user, c->h_red, (int64_t)count, f64 ? 1 : 0);
        if (rc != 0) return fail(c, DPMM_ECOMM, "host all-reduce callback failed (code " + std::to_string(rc) + ")");
        HIPCHK(c, launch_copy_bytes(dbuf, c->h_red, bytes, c->stream));
    }
    if (c->opt_timing & 4) HIPCHK(c, hipEventRecord(c->ev_comm[2 * kind + 1], c->stream));
    c->have_comm_ev[kind] = (c->opt_timing & 4) != 0;
    c->comm_bytes[kind] = (int64_t)count * 8;
    ++c->comm_calls;
    return DPMM_OK;
}
static void watchdog_stop(dpmm_ctx *c) {
    if (!c->watchdog) return;
    { std::lock_guard<std::mutex> lk(c->wd_mu); c->wd_stop = true; }
    c->wd_cv.notify_all();
    c->watchdog->join();
    delete c->watchdog;
    c->watchdog = nullptr;
    c->wd_stop = false;
}
static void watchdog_start(dpmm_ctx *c) {
    if (c->watchdog || !rccl().CommAbort) return;
    c->watchdog = new std::thread([c]() {
        std::unique_lock<std::mutex> lk(c->wd_mu);
        while (!c->wd_stop) {
            c->wd_cv.wait_for(lk, std::chrono::milliseconds(20));
            if (c->wd_stop) break;
            const int64_t since = c->wd_wait_since.load(std::memory_order_relaxed);
            const int limit = c->comm_timeout_ms;
            if (since != 0 && limit > 0 && steady_ms() - since > limit && !c->comm_aborted.load()) {
                c->comm_aborted.store(true);
                rccl().CommAbort(c->comm);          // the collective's kernel returns, the blocked host call with it
            }
        }
    });
}
static void comm_release(dpmm_ctx *c) {
    watchdog_stop(c);
    if (c->comm) { if (!c->comm_aborted.load()) rccl().CommDestroy(c->comm); c->comm = nullptr; }      // (ncclCommAbort released an aborted one)
    c->comm_aborted.store(false);
    c->coll_pending = false; c->abort_reported = false;
    c->host_fn = nullptr; c->host_user = nullptr;
    c->world = 1; c->rank = 0;
    c->have_comm_ev[0] = c->have_comm_ev[1] = false;
    c->comm_bytes[0] = c->comm_bytes[1] = 0; c->comm_calls = 0;
}

#pragma GCC visibility push(default)
extern "C" {

int dpmm_abi_version(void) { return DPMM_ABI_VERSION; }

const char *dpmm_last_error(const dpmm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static void free_params(dpmm_ctx *c) {
    hipFree(c->d_raw); hipFree(c->d_mu); hipFree(c->d_Rp); hipFree(c->d_mup); hipFree(c->d_cst);
    hipFree(c->d_ccache); c->d_ccache = nullptr;
    hipFree(c->d_red); c->d_red = nullptr;
    hipFree(c->d_scratch); hipFree(c->d_slabs); hipFree(c->d_out); hipFree(c->d_Lp16); hipFree(c->d_tdf); hipFree(c->d_lam); hipFree(c->d_mdist); hipFree(c->d_tail); hipFree(c->d_refb_big);
    hipFree(c->d_raw2); hipFree(c->d_Rp2); hipFree(c->d_Lp16_2); c->d_raw2 = c->d_Rp2 = nullptr; c->d_Lp16_2 = nullptr; c->mspec_valid = false;
    c->d_Lp16 = nullptr; c->d_tdf = nullptr; c->d_lam = nullptr; c->d_mdist = nullptr; c->d_tail = nullptr; c->d_refb_big = nullptr; c->have_refb_big = false;
    c->d_raw = c->d_mu = c->d_Rp = c->d_mup = c->d_cst = c->d_scratch = nullptr;
    c->d_slabs = c->d_out = nullptr;
}

// (re)allocate everything whose size depends on the number of clusters
static int ensure_capacity(dpmm_ctx *c, int K) {
    if (K <= c->Kcap) return DPMM_OK;
    int cap = std::max(8, c->Kcap);
    while (cap < K) cap *= 2;
    cap = std::min(cap, DPMM_MAX_CLUSTERS);
    HIPCHK(c, sync_stream(c, c->stream));
    free_params(c);
    const size_t D = (size_t)c->D;
    if (c->prior == DPMM_PRIOR_NIW) {
        const size_t NP = (size_t)c->NB * (c->NB + 1) / 2;
        HIPCHK(c, hipMalloc(&c->d_raw, sizeof(float) * 3 * cap * D * D));
        HIPCHK(c, hipMalloc(&c->d_mu, sizeof(float) * 3 * cap * D));
        HIPCHK(c, hipMalloc(&c->d_Rp, sizeof(float) * 3 * cap * NP * 256));
        HIPCHK(c, hipMalloc(&c->d_mup, sizeof(float) * 3 * cap * 16 * c->NB));
        HIPCHK(c, hipMalloc(&c->d_lam, sizeof(float) * cap));
        if (c->NB == 8 || c->NB == 16) HIPCHK(c, hipMalloc(&c->d_refb_big, sizeof(uint32_t) * cap * niw_refb_big_words(c->NB)));
        HIPCHK(c, hipMalloc(&c->d_tail, sizeof(float) * (c->NB == 4 ? niw_tail_floats(cap) : 16 * (cap + 2) + 16 * cap + (size_t)REFB_WORDS * cap)));      // pair records | per-cluster ball records | bf16 images of the reference bracket | (NB = 4) three-plane images + offsets of niw_lean.hip
        HIPCHK(c, hipMalloc(&c->d_mdist, sizeof(float) * (size_t)cap * cap));
    } else {
        const size_t NT = (size_t)(c->ldx + 15) / 16, NRB = (size_t)(3 * cap + 15) / 16;
        HIPCHK(c, hipMalloc(&c->d_raw, sizeof(float) * 3 * cap * (size_t)c->ldx));
        HIPCHK(c, hipMalloc(&c->d_Rp, sizeof(float) * NRB * NT * 256));
        HIPCHK(c, hipMalloc(&c->d_Lp16, sizeof(uint32_t) * std::max(mult_pack_bf16_words(3 * cap, c->ldx), mult_pack_u8_words(3 * cap, (c->D + 127) / 128 * 128))));
        HIPCHK(c, hipMalloc(&c->d_raw2, sizeof(float) * 3 * cap * (size_t)c->ldx));
        HIPCHK(c, hipMalloc(&c->d_Rp2, sizeof(float) * NRB * NT * 256));
        HIPCHK(c, hipMalloc(&c->d_Lp16_2, sizeof(uint32_t) * std::max(mult_pack_bf16_words(3 * cap, c->ldx), mult_pack_u8_words(3 * cap, (c->D + 127) / 128 * 128))));
    }
    HIPCHK(c, hipMalloc(&c->d_cst, sizeof(float) * 3 * cap));
    HIPCHK(c, hipMalloc(&c->d_tdf, sizeof(float) * 6 * cap));
    // sweep scratch: one a_k row set per resident workgroup; the Multinomial kernel keeps all 3K rows
    const int rows = (c->prior == DPMM_PRIOR_NIW) ? cap : 3 * cap;
    c->scratch_stride = (int64_t)c->sweep_grid * c->tile;
    HIPCHK(c, hipMalloc(&c->d_scratch, sizeof(float) * (size_t)rows * (size_t)c->scratch_stride));
    // statistics
    c->max_items = (int)((c->n + c->chunk - 1) / c->chunk) + 2 * cap;
    // NIW: one slab per SLOT (a workgroup of the statistics kernel, or a bin whose first item lies inside a workgroup's range); Multinomial: per item
    const size_t nslabs = c->prior == DPMM_PRIOR_NIW ? (size_t)NIW_STATS_MAX_GROUPS + 2 * (size_t)cap : (size_t)c->max_items;
    HIPCHK(c, hipMalloc(&c->d_slabs, sizeof(double) * nslabs * (size_t)c->slab_stride));
    HIPCHK(c, hipMalloc(&c->d_out, sizeof(double) * 2 * cap * (size_t)c->packed_stride + DPMM_MAX_CLUSTERS + 64));   // rows | bad-cluster flags
    HIPCHK(c, hipMalloc(&c->d_red, sizeof(double) * 3 * (size_t)cap * (size_t)c->packed_stride));
    HIPCHK(c, hipMalloc(&c->d_ccache, sizeof(double) * cap * (size_t)c->packed_stride));
    HIPCHK(c, hipMemsetAsync(c->d_ccache, 0, sizeof(double) * cap * (size_t)c->packed_stride, c->stream));      // (on the stream its readers run on)
    c->cache_force = true;
    c->Kcap = cap;
    return DPMM_OK;
}

// words of ONE tile list of the lean sweep: count + (position, count) per tile; at most ceil(n / 64) + NIW_LEAN_MAX_BINS tiles (bin-aligned)
static size_t hard_list_words(int64_t n) { return 4 + 2 * ((size_t)((n + 63) / 64) + NIW_LEAN_MAX_BINS + 2); }

int dpmm_create(dpmm_ctx **out, int prior_kind, int D, int64_t n_local, int64_t first_index, int device, uint64_t seed) {
    if (!out) return fail(nullptr, DPMM_EINVAL, "ctx out pointer is null");
    *out = nullptr;
    if (prior_kind != DPMM_PRIOR_NIW && prior_kind != DPMM_PRIOR_MULT) return fail(nullptr, DPMM_EINVAL, "unknown prior kind");
    if (D < 1 || n_local < 0 || first_index < 0) return fail(nullptr, DPMM_EINVAL, "bad D / n_local / first_index");
    if (n_local > 2000000000LL) return fail(nullptr, DPMM_ELIMIT, "n_local exceeds the Int32 point index of this build");
    if (prior_kind == DPMM_PRIOR_NIW && D > DPMM_MAX_DIM_NIW) return fail(nullptr, DPMM_ELIMIT, "NIW: D > DPMM_MAX_DIM_NIW");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, DPMM_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, DPMM_ENODEVICE, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, DPMM_ENODEVICE, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, DPMM_ENODEVICE, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, DPMM_ENODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");

    dpmm_ctx *c = new dpmm_ctx();
    c->prior = prior_kind; c->D = D; c->n = n_local; c->first = first_index; c->device = device; c->seed = seed;
    c->ldx = (D + 3) / 4 * 4;
    // Multinomial rows are streamed in 128-byte pieces per point: pad the row to a multiple of 32 floats so that no piece straddles
    // two cache lines (D = 1000: rows of 4000 B made every piece touch two lines, HBM reads 1.2-1.3 x the algorithmic bytes)
    if (prior_kind == DPMM_PRIOR_MULT && D >= 32) c->ldx = (D + 31) / 32 * 32;
    c->cus = prop.multiProcessorCount;
    auto bail = [&](int code) { std::string m = c->err; dpmm_destroy(c); g_create_error = m; return code; };
#define CHK_CREATE(expr)                                                                  \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e__); return bail(DPMM_EHIP); } \
    } while (0)
    CHK_CREATE(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &e : c->ev) CHK_CREATE(hipEventCreate(&e));
    if (prior_kind == DPMM_PRIOR_NIW) {
        c->NB = nb_for_dim(D);
        c->tile = niw_tile_points(c->NB);
        c->slab_stride = niw_slab_stride(D);
        c->packed_stride = 1 + (int64_t)D + (int64_t)D * (D + 1) / 2;
        c->sweep_grid = c->cus * niw_occupancy(c->NB);
        CHK_CREATE(hipMalloc(&c->d_row_off, sizeof(int32_t) * (size_t)c->packed_stride));
        CHK_CREATE(hipMalloc(&c->d_inv_off, sizeof(int32_t) * (size_t)c->slab_stride));
        CHK_CREATE(launch_niw_row_offsets(c->d_row_off, c->d_inv_off, D, c->packed_stride, c->stream));
    } else {
        c->tile = mult_tile_points();
        c->slab_stride = mult_slab_stride(D);
        c->packed_stride = 1 + (int64_t)D;
        c->sweep_grid = c->cus * 2;
    }
    c->ntiles = (n_local + c->tile - 1) / c->tile;
    if (c->ntiles < c->sweep_grid) c->sweep_grid = (int)std::max<int64_t>(1, c->ntiles);
    c->sweep_grid_max = c->sweep_grid;
    // statistics work items: ~4 per resident wave slot (2 waves/SIMD) so the last round is not mostly empty
    // work items of the statistics pass: the NIW kernel hands every workgroup a contiguous RANGE of items, so finer items
    // only improve the balance (ceil(items / groups) granularity) -- 16384 at D <= 64 (1.05 -> 0.85 ms at N = 1e7); slabs
    // are allocated per item, which is why the large-D kernels (280 KB per slab at D = 256) stay at 8192
    const int64_t target_items = (c->prior == DPMM_PRIOR_NIW && c->D <= 64) ? 16384 : 8192;
    // NIW: the statistics kernel balances its workgroups' ranges to ONE item and slabs are per slot, not per item -- items only set the
    // granularity of the balance: 32 points (8 k-steps) for the one-wave D <= 64 kernels, 256 for the multi-panel kernels (16-point batches).
    // Multinomial: one workgroup and one slab per item: 8192 items.
    const int64_t min_chunk = (c->prior == DPMM_PRIOR_NIW && c->D <= 64) ? 32 : 256;
    const int64_t target_items2 = (c->prior == DPMM_PRIOR_NIW && c->D <= 64) ? 65536 : target_items;
    c->chunk = (int)std::max<int64_t>(min_chunk, ((n_local + target_items2 - 1) / target_items2 + 3) / 4 * 4);
    const size_t nalloc = (size_t)std::max<int64_t>(n_local, 1);
    CHK_CREATE(hipMalloc(&c->dX, sizeof(float) * nalloc * (size_t)c->ldx));
    CHK_CREATE(hipMalloc(&c->dbins, sizeof(int32_t) * nalloc));
    // sort tiles: 512 points per sorting wave below 4e6 points (the tile kernels are one-wave latency chains: at the 8-GPU shard size four
    // times as many waves of a quarter of the trips each), 2048 above; the tables are sized for whichever is in use (DPMM_OPT_SORT_TILE
    // may switch while the shard is small enough for the small tile's tables)
    c->sb.tile = n_local <= 4000000 ? SORT_TILE_SMALL : SORT_TILE;
    const int alloc_tile = c->sb.tile;           // (DPMM_OPT_SORT_TILE re-allocates when it selects the smaller tile: at N = 1e7 the small tile's tables are 320 MB, these 80 MB)
    c->sort_tile_min = alloc_tile;
    c->nt_sort = (int)((n_local + alloc_tile - 1) / alloc_tile);
    const size_t nbmax = 2 * DPMM_MAX_CLUSTERS;
    CHK_CREATE(hipMalloc(&c->sb.tile_hist, sizeof(int32_t) * nbmax * (size_t)std::max(1, c->nt_sort)));
    CHK_CREATE(hipMalloc(&c->sb.tile_cnt, sizeof(int32_t) * nbmax * (size_t)std::max(1, c->nt_sort)));
    CHK_CREATE(hipMalloc(&c->sb.tile_spec, sizeof(int32_t) * (size_t)STEP_SPEC_MAX_BINS * (size_t)std::max(1, c->nt_sort)));
    CHK_CREATE(hipMalloc(&c->sb.spec_bins, sizeof(int32_t) * nalloc));
    CHK_CREATE(hipMalloc(&c->sb.fast_total, sizeof(int32_t) * nbmax * FAST_TOTAL_STRIDE));
    CHK_CREATE(hipMemsetAsync(c->sb.fast_total, 0, sizeof(int32_t) * nbmax * FAST_TOTAL_STRIDE, c->stream));
    CHK_CREATE(hipMalloc(&c->sb.ticket, sizeof(unsigned)));
    CHK_CREATE(hipMemsetAsync(c->sb.ticket, 0, sizeof(unsigned), c->stream));
    CHK_CREATE(hipMalloc(&c->sb.prev_lab, sizeof(uint16_t) * (((size_t)nalloc + SORT_TILE - 1) / SORT_TILE * SORT_TILE)));
    CHK_CREATE(hipMemsetAsync(c->sb.prev_lab, 0xFF, sizeof(uint16_t) * (((size_t)nalloc + SORT_TILE - 1) / SORT_TILE * SORT_TILE), c->stream));
    CHK_CREATE(hipMalloc(&c->sb.cdirty, DPMM_MAX_CLUSTERS + 8));
    CHK_CREATE(hipMemsetAsync(c->sb.cdirty, 1, DPMM_MAX_CLUSTERS + 8, c->stream));
    CHK_CREATE(hipMalloc(&c->sb.cmode, DPMM_MAX_CLUSTERS));
    CHK_CREATE(hipMemsetAsync(c->sb.cmode, 0, DPMM_MAX_CLUSTERS, c->stream));
    CHK_CREATE(hipMalloc(&c->sb.bin_total, sizeof(int32_t) * nbmax));
    CHK_CREATE(hipMalloc(&c->sb.bin_start, sizeof(int32_t) * (nbmax + 1)));
    CHK_CREATE(hipMalloc(&c->sb.item_start, sizeof(int32_t) * (nbmax + 1)));
    CHK_CREATE(hipMalloc(&c->sb.perm, sizeof(int32_t) * nalloc));
    CHK_CREATE(hipMalloc(&c->sb.bin_sel, nbmax));
    c->sel_capacity = (int)nbmax;
    CHK_CREATE(hipMalloc(&c->sb.perm_total, sizeof(int32_t)));
    CHK_CREATE(hipMalloc(&c->d_small, sizeof(int32_t) * 4 * DPMM_MAX_CLUSTERS));
    CHK_CREATE(hipMalloc(&c->d_counts64, sizeof(long long) * 2 * DPMM_MAX_CLUSTERS));
    CHK_CREATE(hipMalloc(&c->d_cside, DPMM_MAX_CLUSTERS));
    CHK_CREATE(hipMemsetAsync(c->d_cside, 0, DPMM_MAX_CLUSTERS, c->stream));
    if (c->prior == DPMM_PRIOR_NIW && c->NB == 4) {
        CHK_CREATE(hipMalloc(&c->d_sp_frag, sizeof(uint32_t) * (size_t)SP_MAXK * SP_FRAG_WORDS));
        CHK_CREATE(hipMalloc(&c->d_sp_cons, sizeof(float) * (size_t)SP_MAXK * SP_CONS_FLOATS));
        // ([8 grid]: two words per wave of the sweep kernel | [4 grid]: tiles the lean kernel settled, per wave)
        // [0, 8 g): two words per wave of the general kernel | [8 g, 12 g): one word per wave of the lean kernel (tiles settled without the direction
        // screen) | [12 g, 20 g): two words per wave of the lean kernel WITH the direction screen (round 6)      (g = sweep_grid_max workgroups of 4 waves)
        CHK_CREATE(hipHostMalloc((void **)&c->h_need, sizeof(uint32_t) * 20 * (size_t)std::max(1, c->sweep_grid_max), hipHostMallocDefault));
        memset(c->h_need, 0, sizeof(uint32_t) * 20 * (size_t)std::max(1, c->sweep_grid_max));
        if (c->NB == 4) {
            const size_t hw = hard_list_words(n_local);
            CHK_CREATE(hipMalloc(&c->d_hard, sizeof(uint32_t) * 2 * hw));
            CHK_CREATE(hipMemsetAsync(c->d_hard, 0, sizeof(uint32_t) * 2 * hw, c->stream));
            CHK_CREATE(hipHostMalloc((void **)&c->h_hard, 64, hipHostMallocDefault));
            memset(c->h_hard, 0, 64);
        }
    }
    CHK_CREATE(hipMalloc(&c->d_work, sizeof(unsigned long long) * (DPMM_WORK_SLOTS + 4 * DPMM_WORK_PER_WAVE * (size_t)std::max(1, c->sweep_grid_max))));
    CHK_CREATE(hipMemsetAsync(c->d_work, 0, sizeof(unsigned long long) * (DPMM_WORK_SLOTS + 4 * DPMM_WORK_PER_WAVE * (size_t)std::max(1, c->sweep_grid_max)), c->stream));
#undef CHK_CREATE
    *out = c;
    return DPMM_OK;
}

int dpmm_destroy(dpmm_ctx *c) {
    if (!c) return DPMM_OK;
    hipSetDevice(c->device);
    if (c->stream && !c->comm_aborted.load()) hipStreamSynchronize(c->stream);
    free_params(c);
    hipFree(c->dX); hipFree(c->dX8); hipFree(c->dbins); hipFree(c->d_gt); hipFree(c->d_cont);
    hipFree(c->sb.tile_hist); hipFree(c->sb.tile_cnt); hipFree(c->sb.tile_spec); hipFree(c->sb.spec_bins); hipFree(c->sb.fast_total); hipFree(c->sb.ticket); hipFree(c->sb.prev_lab); hipFree(c->sb.cdirty); hipFree(c->sb.cmode); hipFree(c->sb.bin_total); hipFree(c->sb.bin_start); hipFree(c->sb.item_start);
    hipFree(c->sb.perm); hipFree(c->sb.bin_sel); hipFree(c->sb.perm_total); hipFree(c->d_small); hipFree(c->d_proj); hipFree(c->d_vals); hipFree(c->d_smart);
    hipFree(c->d_m0); hipFree(c->d_psi_lo); hipFree(c->d_pairs);
    for (int i = 0; i < 2; ++i) { hipFree(c->d_Y[i]); hipFree(c->d_ld_sigma[i]); hipFree(c->d_mu_draw[i]); }
    hipFree(c->ma.fac); hipFree(c->ma.mean); hipFree(c->ma.kap); hipFree(c->ma.nu); hipFree(c->ma.rows_store);
    if (c->stream2) { hipStreamSynchronize(c->stream2); hipStreamDestroy(c->stream2); }
    for (auto &e : c->ev_part) if (e) hipEventDestroy(e);
    if (c->ev_master) hipEventDestroy(c->ev_master);
    if (c->ev_spec) hipEventDestroy(c->ev_spec);
    if (c->ev_noise) hipEventDestroy(c->ev_noise);
    if (c->ev_pairs) hipEventDestroy(c->ev_pairs);
    if (c->ev_rows) hipEventDestroy(c->ev_rows);
    hipFree(c->d_apairs);
    if (c->h_apairs) hipHostFree(c->h_apairs);
    for (auto &b : c->h_apairs_list) if (b) hipHostFree(b);
    hipFree(c->d_jobs); hipFree(c->d_dslots);
    for (int i = 0; i < 4; ++i) if (c->h_list[i]) hipHostFree(c->h_list[i]);
    if (c->h_master) hipHostFree(c->h_master);
    if (c->h_draw) hipHostFree(c->h_draw);
    hipFree(c->d_malpha); hipFree(c->d_mpairs);
    if (c->h_marg) hipHostFree(c->h_marg);
    hipFree(c->d_sp_frag); hipFree(c->d_sp_cons); hipFree(c->d_brk_flag); hipFree(c->d_brk_aref);
    if (c->h_need) hipHostFree(c->h_need);
    if (c->h_hard) hipHostFree(c->h_hard);
    hipFree(c->d_hard);
    hipFree(c->d_counts64); hipFree(c->d_cside); hipFree(c->d_row_off); hipFree(c->d_inv_off); hipFree(c->d_work); hipFree(c->d_par);
    comm_release(c);
    if (c->h_red) hipHostFree(c->h_red);
    for (auto &e : c->ev_comm) if (e) hipEventDestroy(e);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->h_par) hipHostFree(c->h_par);
    if (c->h_out) hipHostFree(c->h_out);
    for (auto &e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return DPMM_OK;
}

// after the points are in place: the Multinomial fast path needs to know whether every count is exact in bf16
static int finish_upload(dpmm_ctx *c) {
    if (c->prior == DPMM_PRIOR_MULT) {
        const bool force_f32 = c->opt_force_f32 != 0;
        const int was_u8 = c->x_u8, was_bf16 = c->x_bf16_exact;
        int *flag = reinterpret_cast<int *>(c->d_small);
        HIPCHK(c, hipMemsetAsync(flag, 0, sizeof(int), c->stream));
        HIPCHK(c, launch_bf16_exact_check(c->dX, c->n * c->ldx, flag, c->stream));
        int h = 1;
        HIPCHK(c, hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, sync_stream(c, c->stream));
        c->x_bf16_exact = (h == 0 && !force_f32) ? 1 : 0;
        c->x_u8 = 0;
        if (c->x_bf16_exact && !c->opt_no_u8 && c->n > 0) {
            // small non-negative integers (bag-of-words counts): keep a lossless byte copy and stream 1 byte per element
            c->ld8 = (c->D + 127) / 128 * 128;
            if (!c->dX8) HIPCHK(c, hipMalloc(&c->dX8, (size_t)c->n * (size_t)c->ld8));
            HIPCHK(c, hipMemsetAsync(flag, 0, sizeof(int), c->stream));
            HIPCHK(c, launch_u8_convert(c->dX, c->ldx, c->D, c->n, c->dX8, c->ld8, flag, c->stream));
            HIPCHK(c, hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, sync_stream(c, c->stream));
            c->x_u8 = (h == 0) ? 1 : 0;
            if (!c->x_u8) { hipFree(c->dX8); c->dX8 = nullptr; }
            // every Multinomial kernel of a byte-path context reads the byte copy (a lossless re-encoding): the Float32 matrix (4 bytes per
            // element, 4 GB at D = 1000, N = 1e6) is dead weight from here on; a new upload allocates it again
            else { HIPCHK(c, hipFree(c->dX)); c->dX = nullptr; }
        }
        // parameters that stay in place across the upload were packed for the OLD points' sweep kernel (byte planes / bf16 planes / Float32
        // fragments): when the new points take another kernel, the images it reads are made from the raw rows now
        if (c->have_params && c->d_raw && (was_u8 != c->x_u8 || was_bf16 != c->x_bf16_exact)) {
            HIPCHK(c, launch_mult_pack(c->d_raw, c->d_Rp, 3 * c->K, c->ldx, c->stream));
            c->rp_current = true;
            if (c->x_u8) HIPCHK(c, launch_mult_pack_u8(c->d_raw, c->d_Lp16, 3 * c->K, c->ldx, c->ld8, c->stream));
            else if (c->x_bf16_exact) HIPCHK(c, launch_mult_pack_bf16(c->d_raw, c->d_Lp16, 3 * c->K, c->ldx, c->stream));
            c->mspec_valid = false;
        }
    }
    HIPCHK(c, sync_stream(c, c->stream));
    // new points: the cached cluster-level statistics (derive_rows_kernel) and the rows a device master would draw from belong to the old ones
    c->cache_force = true;
    c->rows_full_K = -1;
    return DPMM_OK;
}

static int ensure_points_buffer(dpmm_ctx *c) {       // (a byte-path Multinomial context gave its Float32 matrix back)
    if (!c->dX) HIPCHK(c, hipMalloc(&c->dX, sizeof(float) * (size_t)std::max<int64_t>(c->n, 1) * (size_t)c->ldx));
    return DPMM_OK;
}
static int upload_common(dpmm_ctx *c, const float *X, int64_t ldx, hipMemcpyKind kind) {
    if (!c) return DPMM_EINVAL;
    if (!X && c->n > 0) return fail(c, DPMM_EINVAL, "X is null");
    if (ldx < c->D) return fail(c, DPMM_EINVAL, "ldx < D");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_points_buffer(c)) return rc;
    if (c->n > 0) {
        if (c->ldx != c->D) HIPCHK(c, hipMemsetAsync(c->dX, 0, sizeof(float) * (size_t)c->n * c->ldx, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(c->dX, sizeof(float) * c->ldx, X, sizeof(float) * ldx, sizeof(float) * c->D, (size_t)c->n, kind, c->stream));
        if (int rc = finish_upload(c)) return rc;
    }
    c->have_points = true;
    c->cache_force = true;          // (also for n == 0; finish_upload does it for every path that moved points)
    c->rows_full_K = -1;
    return DPMM_OK;
}

int dpmm_upload_points(dpmm_ctx *c, const float *X, int64_t ldx) { return upload_common(c, X, ldx, hipMemcpyHostToDevice); }
int dpmm_upload_points_device(dpmm_ctx *c, const float *dX, int64_t ldx) { return upload_common(c, dX, ldx, hipMemcpyDeviceToDevice); }

int dpmm_upload_points_npy(dpmm_ctx *c, const void *rows, int is_f64, int64_t ld, int nan_to_zero) {
    if (!c) return DPMM_EINVAL;
    if (!rows && c->n > 0) return fail(c, DPMM_EINVAL, "rows is null");
    if (ld < c->D) return fail(c, DPMM_EINVAL, "ld < D");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_points_buffer(c)) return rc;
    if (c->n > 0) {
        if (c->ldx != c->D) HIPCHK(c, hipMemsetAsync(c->dX, 0, sizeof(float) * (size_t)c->n * c->ldx, c->stream));
        const size_t esz = is_f64 ? sizeof(double) : sizeof(float);
        const int64_t chunk_rows = std::max<int64_t>(1, std::min<int64_t>(c->n, ((int64_t)128 << 20) / (int64_t)(esz * (size_t)ld)));
        void *tmp = nullptr;
        HIPCHK(c, hipMalloc(&tmp, esz * (size_t)chunk_rows * (size_t)ld));
        int rc = DPMM_OK;
        for (int64_t r0 = 0; r0 < c->n && rc == DPMM_OK; r0 += chunk_rows) {
            const int64_t nr = std::min(chunk_rows, c->n - r0);
            hipError_t e = hipMemcpyAsync(tmp, (const char *)rows + esz * (size_t)r0 * (size_t)ld, esz * (size_t)nr * (size_t)ld, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = launch_ingest_rows(c->dX + (size_t)r0 * c->ldx, c->ldx, tmp, is_f64, ld, nr, c->D, nan_to_zero, c->stream);
            if (e == hipSuccess) e = sync_stream(c, c->stream);       // tmp is reused by the next chunk
            if (e != hipSuccess) { c->err = std::string("dpmm_upload_points_npy: ") + hipGetErrorString(e); rc = DPMM_EHIP; }
        }
        hipFree(tmp);
        if (rc != DPMM_OK) return rc;
        if (int rc2 = finish_upload(c)) return rc2;
    }
    c->have_points = true;
    c->cache_force = true;
    c->rows_full_K = -1;
    return DPMM_OK;
}

int dpmm_init_labels(dpmm_ctx *c, int init_clusters, uint32_t epoch) {
    if (!c) return DPMM_EINVAL;
    if (init_clusters < 1 || init_clusters > DPMM_MAX_CLUSTERS) return fail(c, DPMM_ELIMIT, "init_clusters out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n > 0) HIPCHK(c, launch_init_labels(c->dbins, c->n, c->first, init_clusters, 0, c->seed, epoch, c->stream));
    c->have_labels = true;
    c->rows_full_K = -1;
    return DPMM_OK;
}

int dpmm_init_labels_from(dpmm_ctx *c, int init_clusters, int first_label, uint32_t epoch) {
    if (!c) return DPMM_EINVAL;
    if (first_label < 1 || init_clusters < 1 || first_label - 1 + init_clusters > DPMM_MAX_CLUSTERS) return fail(c, DPMM_ELIMIT, "init_clusters / first_label out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n > 0) HIPCHK(c, launch_init_labels(c->dbins, c->n, c->first, init_clusters, first_label - 1, c->seed, epoch, c->stream));
    c->have_labels = true;
    c->rows_full_K = -1;
    return DPMM_OK;
}

int dpmm_set_labels(dpmm_ctx *c, const int64_t *labels, const int64_t *sub) {
    if (!c) return DPMM_EINVAL;
    if (!c->have_labels && (!labels || !sub)) return fail(c, DPMM_ESTATE, "first dpmm_set_labels must provide both vectors");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) { c->have_labels = true; return DPMM_OK; }
    int64_t *tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, sizeof(int64_t) * 2 * (size_t)c->n));
    int rc = DPMM_OK;
    hipError_t e = hipSuccess;
    if (labels) e = hipMemcpyAsync(tmp, labels, sizeof(int64_t) * c->n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && sub) e = hipMemcpyAsync(tmp + c->n, sub, sizeof(int64_t) * c->n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_bins_from_i64(c->dbins, labels ? tmp : nullptr, sub ? tmp + c->n : nullptr, c->n, c->stream);
    if (e == hipSuccess) e = sync_stream(c, c->stream);
    hipFree(tmp);
    if (e != hipSuccess) { c->err = std::string("dpmm_set_labels: ") + hipGetErrorString(e); rc = DPMM_EHIP; }
    else c->have_labels = true;
    c->rows_full_K = -1;
    return rc;
}

int dpmm_get_labels(dpmm_ctx *c, int64_t *labels, int64_t *sub) {
    if (!c) return DPMM_EINVAL;
    if (!c->have_labels) return fail(c, DPMM_ESTATE, "labels not initialised");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) return DPMM_OK;
    int64_t *tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, sizeof(int64_t) * 2 * (size_t)c->n));
    hipError_t e = launch_bins_to_i64(c->dbins, tmp, tmp + c->n, c->n, c->stream);
    if (e == hipSuccess && labels) e = hipMemcpyAsync(labels, tmp, sizeof(int64_t) * c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && sub) e = hipMemcpyAsync(sub, tmp + c->n, sizeof(int64_t) * c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = sync_stream(c, c->stream);
    hipFree(tmp);
    if (e != hipSuccess) { c->err = std::string("dpmm_get_labels: ") + hipGetErrorString(e); return DPMM_EHIP; }
    return DPMM_OK;
}

static int check_K(dpmm_ctx *c, int K) {
    if (K < 1) return fail(c, DPMM_EINVAL, "K < 1");
    if (K > DPMM_MAX_CLUSTERS) return fail(c, DPMM_ELIMIT, "K > DPMM_MAX_CLUSTERS");
    return DPMM_OK;
}

// ---- parameter staging: slot-indexed rows in pinned, GPU-addressable host memory ------------------------------------------
// Layout for `slots` clusters (Float32 unless noted):
//   NIW : mu [3 slots][D] | R [3 slots][D(D+1)/2] (packed upper triangle) | logdet [3 slots] | lr [slots][2] | w [slots] | cst [3 slots] | slot map Int32 [slots]
//   MULT:                   logp [3 slots][D]                   | lr [slots][2] | w [slots] | cst [3 slots] | slot map Int32 [slots]
struct ParLayout {
    size_t mu = 0, mat = 0, logdet = 0, lr = 0, w = 0, cst = 0, slot = 0, bytes = 0;
};
static ParLayout par_layout(const dpmm_ctx *c, int slots) {
    ParLayout L;
    const size_t D = (size_t)c->D, S = (size_t)slots;
    size_t o = 0;
    if (c->prior == DPMM_PRIOR_NIW) {
        // cst | mu | mat are contiguous (every offset a multiple of 16 bytes: slots is a multiple of 8): ONE copy takes them to the device
        L.cst = o; o += sizeof(float) * 3 * S;
        L.mu = o; o += sizeof(float) * 3 * S * D;
        L.mat = o; o += sizeof(float) * 3 * S * (D * (D + 1) / 2);
        L.logdet = o; o += sizeof(float) * 3 * S;
    } else {
        L.mat = o; o += sizeof(float) * 3 * S * D;
    }
    L.lr = o; o += sizeof(float) * 2 * S;
    L.w = o; o += sizeof(float) * S;
    if (c->prior != DPMM_PRIOR_NIW) { L.cst = o; o += sizeof(float) * 3 * S; }
    L.slot = o; o += sizeof(int32_t) * S;
    L.bytes = (o + 255) & ~(size_t)255;
    return L;
}

int dpmm_params_staging(dpmm_ctx *c, int slots, float **mu, float **mat, float **logdet, float **lr, float **w, int32_t **slot_of_cluster) {
    if (!c) return DPMM_EINVAL;
    if (slots < 1 || slots > DPMM_MAX_CLUSTERS) return fail(c, DPMM_ELIMIT, "slots out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (slots > c->par_slots) {
        int ns = std::max(8, c->par_slots);
        while (ns < slots) ns *= 2;
        ns = std::min(ns, DPMM_MAX_CLUSTERS);
        const ParLayout N = par_layout(c, ns);
        char *nb = nullptr;
        HIPCHK(c, sync_stream(c, c->stream));      // a pack kernel may still read the old buffer
        HIPCHK(c, hipHostMalloc((void **)&nb, N.bytes, hipHostMallocDefault));
        memset(nb, 0, N.bytes);
        if (c->h_par) {                                  // contents are preserved (rows are slot-indexed: same offsets inside a region)
            const ParLayout O = par_layout(c, c->par_slots);
            const size_t D = (size_t)c->D, S = (size_t)c->par_slots;
            if (c->prior == DPMM_PRIOR_NIW) {
                memcpy(nb + N.mu, c->h_par + O.mu, sizeof(float) * 3 * S * D);
                memcpy(nb + N.mat, c->h_par + O.mat, sizeof(float) * 3 * S * (D * (D + 1) / 2));
                memcpy(nb + N.logdet, c->h_par + O.logdet, sizeof(float) * 3 * S);
            } else {
                memcpy(nb + N.mat, c->h_par + O.mat, sizeof(float) * 3 * S * D);
            }
            memcpy(nb + N.lr, c->h_par + O.lr, sizeof(float) * 2 * S);
            memcpy(nb + N.w, c->h_par + O.w, sizeof(float) * S);
            memcpy(nb + N.slot, c->h_par + O.slot, sizeof(int32_t) * S);
            hipHostFree(c->h_par);
        }
        c->h_par = nb; c->h_par_bytes = N.bytes; c->par_slots = ns;
    }
    const ParLayout L = par_layout(c, c->par_slots);
    if (mu) *mu = c->prior == DPMM_PRIOR_NIW ? reinterpret_cast<float *>(c->h_par + L.mu) : nullptr;
    if (mat) *mat = reinterpret_cast<float *>(c->h_par + L.mat);
    if (logdet) *logdet = c->prior == DPMM_PRIOR_NIW ? reinterpret_cast<float *>(c->h_par + L.logdet) : nullptr;
    if (lr) *lr = reinterpret_cast<float *>(c->h_par + L.lr);
    if (w) *w = reinterpret_cast<float *>(c->h_par + L.w);
    if (slot_of_cluster) *slot_of_cluster = reinterpret_cast<int32_t *>(c->h_par + L.slot);
    return DPMM_OK;
}

static int direction_tables(dpmm_ctx *c, int K, bool b3_done, bool pb_done = false);

int dpmm_commit_params(dpmm_ctx *c, int K) {
    if (!c) return DPMM_EINVAL;
    if (int rc = check_K(c, K)) return rc;
    if (!c->h_par || K > c->par_slots) return fail(c, DPMM_ESTATE, "dpmm_commit_params: call dpmm_params_staging(slots >= K) first");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_capacity(c, K)) return rc;
    const ParLayout L = par_layout(c, c->par_slots);
    const float *hmu = reinterpret_cast<const float *>(c->h_par + L.mu), *hmat = reinterpret_cast<const float *>(c->h_par + L.mat);   // (NIW: re-pointed to the device copy below)
    const float *hld = reinterpret_cast<const float *>(c->h_par + L.logdet), *hlr = reinterpret_cast<const float *>(c->h_par + L.lr);
    const float *hw = reinterpret_cast<const float *>(c->h_par + L.w);
    float *hcst = reinterpret_cast<float *>(c->h_par + L.cst);
    const int32_t *hslot = reinterpret_cast<const int32_t *>(c->h_par + L.slot);
    const bool niw = c->prior == DPMM_PRIOR_NIW;
    for (int k = 0; k < K; ++k) {
        const int sl = hslot[k];
        if (sl < 0 || sl >= c->par_slots) return fail(c, DPMM_EINVAL, "slot_of_cluster entry out of range");
        // cst[3k] = -logdet/2 + log w_k ; cst[3k+1+s] = -logdet/2 + log lr_w[k][s]   (Float32, as the workers of the reference add them)
        hcst[3 * k] = (niw ? -0.5f * hld[3 * sl] : 0.f) + logf(hw[k]);
        hcst[3 * k + 1] = (niw ? -0.5f * hld[3 * sl + 1] : 0.f) + logf(hlr[2 * k]);
        hcst[3 * k + 2] = (niw ? -0.5f * hld[3 * sl + 2] : 0.f) + logf(hlr[2 * k + 1]);
    }
    // the kernels below read the staging buffer in place (no copy-engine transfer); the host may touch it again once a
    // blocking call (dpmm_step_stats, dpmm_suffstats_*, dpmm_sync) has returned
    if (niw) {
        // cst | mu | R of the slots in use: ONE streaming copy over the host link (1 KiB requests), then the gather into the fragment
        // images reads HBM.  (Letting the pack kernel gather from the pinned buffer directly cost 0.9 ms at D = 256 -- 16-byte
        // requests over PCIe -- against 0.25 ms for the bulk copy.)
        int top = 0;
        for (int k = 0; k < K; ++k) top = std::max(top, hslot[k] + 1);
        const size_t D = (size_t)c->D, T = D * (D + 1) / 2;
        const size_t img_bytes = (L.mat - L.cst) + sizeof(float) * 3 * (size_t)top * T;
        const size_t cap = (L.logdet - L.cst) + 1024;
        if (cap > c->d_par_bytes) {
            HIPCHK(c, sync_stream(c, c->stream));
            hipFree(c->d_par); c->d_par = nullptr;
            HIPCHK(c, hipMalloc(&c->d_par, cap));
            c->d_par_bytes = cap;
        }
        HIPCHK(c, launch_copy_bytes16(c->d_par, hcst, img_bytes, c->stream));
        const float *dcst = reinterpret_cast<const float *>(c->d_par);
        hmu = reinterpret_cast<const float *>(c->d_par + (L.mu - L.cst));
        hmat = reinterpret_cast<const float *>(c->d_par + (L.mat - L.cst));
        c->have_tail = c->opt_tail && c->D >= 4 && c->D % 4 == 0 && K > 1;
        HIPCHK(c, launch_niw_pack(hmat, hmu, c->d_Rp, c->d_mup, c->D, c->NB, 3 * K, c->have_tail ? c->d_tail : nullptr, dcst, hslot, c->d_cst,
                                  c->d_work, c->stream));
        c->work_zeroed = true;
        c->have_screen_prep = false;
        if (c->D > 16 && c->D <= 64 && K > 2) {
            // the vectorised far mask (64 clusters per step) pays once there are many clusters, or when the VALU tail
            // screen is unavailable (D not a multiple of 4); DPMM_OPT_PRESCREEN forces it off / on
            const bool want = c->opt_prescreen >= 0 ? c->opt_prescreen != 0 : (!c->have_tail || K > 64);
            if (want) {
                HIPCHK(c, launch_niw_screen_prep(hmat, hmu, c->D, K, c->d_lam, c->d_mdist, hslot, c->stream));
                c->have_screen_prep = true;
            }
        }
        c->predictive = false;          // (before the tables: direction_tables decides the three-plane images from it -- the first parameter set behind a
                                        //  dpmm_set_predictive_* call used to get none and drew one sweep's sub-labels with the Float32 chain: ADVICE r5)
        if (int rc = direction_tables(c, K, false)) return rc;
    } else {
        HIPCHK(c, launch_copy_bytes(c->d_cst, hcst, sizeof(float) * 3 * K, c->stream));
        HIPCHK(c, launch_gather_rows(c->d_raw, c->ldx, hmat, c->D, hslot, 3 * K, c->D, c->stream));
        HIPCHK(c, launch_mult_pack(c->d_raw, c->d_Rp, 3 * K, c->ldx, c->stream));
        c->rp_current = true;
        if (c->x_u8) HIPCHK(c, launch_mult_pack_u8(c->d_raw, c->d_Lp16, 3 * K, c->ldx, c->ld8, c->stream));
        else if (c->x_bf16_exact) HIPCHK(c, launch_mult_pack_bf16(c->d_raw, c->d_Lp16, 3 * K, c->ldx, c->stream));
    }
    c->K = K;
    c->have_params = true;
    c->predictive = false;
    c->draws_on_device = false;
    return DPMM_OK;
}

// Direction screen: the tables of a new parameter set (one small kernel right behind the pack kernels, on the ctx stream).  Built only while
// it pays -- automatic mode: the tiles of the previous sweep kept eight or more candidates behind the 4-row tests on average (the waves
// write their counts into pinned memory; that sweep has been waited for by whoever brings new parameters), with hysteresis: on from 8 per
// tile, off below 4.  While it is on, the sweeps run the screen in front of the 4-row pair tests; the first sweep, every 32nd and the one
// after a change of K keep the usual order and count.
static bool want_pair_ball(const dpmm_ctx *c, int K) { return c->opt_pair_ball && c->opt_ball && K >= 2 && K <= PB_MAXK; }      // (with the images: want_b3_images)
static bool want_b3_images(const dpmm_ctx *c) { return c->prior == DPMM_PRIOR_NIW && c->NB == 4 && c->opt_b3 && c->have_tail && !c->predictive; }
static int direction_tables(dpmm_ctx *c, int K, bool b3_done, bool pb_done) {
    c->sp_ready = false;
    c->have_refb_big = false;
    c->have_b3 = false;
    if (want_b3_images(c)) {
        // the sub-cluster factors' bf16 planes + offsets (niw_lean.hip) -- b3_done: the hand-over launch wrote them (niw_master_pack_roles_kernel)
        // ... and the lean kernel's pair-ball table (same life as the images; one more row of workgroups of the same launch -- or a role of the hand-over launch)
        const bool pb = want_pair_ball(c, K);
        const int what = (b3_done ? 0 : 1) | ((pb && !pb_done) ? 2 : 0);
        if (what) HIPCHK(c, launch_niw_b3_pack(c->d_Rp, c->d_mup, K, c->d_tail, what, c->stream));
        c->have_b3 = true;
        c->have_pb = pb;
    } else c->have_pb = false;
    if (c->prior == DPMM_PRIOR_NIW && (c->NB == 8 || c->NB == 16) && c->d_refb_big && c->opt_bracket && c->have_tail && K > 1) {
        HIPCHK(c, launch_niw_refb_big(c->d_Rp, c->NB, K, c->d_refb_big, c->stream));      // D = 128, 256: the reference bracket's images
        c->have_refb_big = true;
    }
    if (c->prior != DPMM_PRIOR_NIW || c->NB != 4 || !c->d_sp_frag || K < 3 || K > SP_MAXK || !c->have_tail || !c->opt_bf16scr || c->opt_margin <= 0.f) return DPMM_OK;
    bool want = c->opt_direction > 0;
    if (c->opt_direction < 0) {
        unsigned long long many = 0, tiles = 0;
        const int nw = 4 * c->sweep_grid;
        unsigned long long many_ub = 0, tiles_ub = 0;        // sweeps that ran the screen in front of the tail pairs: counted there, an upper bound
        unsigned long long given = 0, removed = 0;            // the last sweep's direction screens, if it ran any
        bool lean_dir = false;                                // ... in the lean kernel
        for (int w = 0; w < nw; ++w) {
            const uint32_t v = c->h_need[2 * w];
            if (v & 0x8000u) { many_ub += v >> 16; tiles_ub += v & 0x7FFFu; } else { many += v >> 16; tiles += v & 0x7FFFu; }
            if (c->sp_last) { const uint32_t y = c->h_need[2 * w + 1]; given += y & 0xFFFFu; removed += y >> 16; }
            tiles += c->h_need[8 * (size_t)c->sweep_grid_max + w];      // tiles niw_lean_kernel settled (no candidate behind the 4-row tests)
            // ... and the tiles it settled WITH the direction screen: their candidates and the screen's yield, in the general kernel's format
            const uint32_t v3 = c->h_need[12 * (size_t)c->sweep_grid_max + 2 * w], y3 = c->h_need[12 * (size_t)c->sweep_grid_max + 2 * w + 1];
            if (v3 & 0x8000u) { many_ub += v3 >> 16; tiles_ub += v3 & 0x7FFFu; } else { many += v3 >> 16; tiles += v3 & 0x7FFFu; }
            given += y3 & 0xFFFFu; removed += y3 >> 16;
            if (y3) lean_dir = true;
        }
        memset(c->h_need, 0, sizeof(uint32_t) * 2 * (size_t)nw);
        memset(c->h_need + 8 * (size_t)c->sweep_grid_max, 0, sizeof(uint32_t) * (size_t)nw);      // (read once: a later launch on a smaller grid, or one that does not count, leaves zeros -- "no tiles")
        memset(c->h_need + 12 * (size_t)c->sweep_grid_max, 0, sizeof(uint32_t) * 2 * (size_t)nw);
        // (break-even measured on the growth run: at 4.3 candidates per tile the screen costs 3 % of the step, at 28 it saves a third)
        if (tiles > 0) c->sp_regime = c->sp_regime ? (many >= tiles * 4) : (many >= tiles * 8);
        else if (tiles_ub > 0 && many_ub < tiles_ub * 4) c->sp_regime = false;        // even the upper bound is below the switch-off level
        // clusters that overlap along every direction (a chain that is still growing: one cluster over several components): the screen is
        // given candidates and removes few -- off for 64 parameter sets, then it may try again
        if ((c->sp_last || lean_dir) && given > 0 && removed * 4 < given) { c->sp_regime = false; c->sp_cooldown = 64; }
        if (c->sp_cooldown > 0) { c->sp_cooldown -= 1; c->sp_regime = false; }
        want = c->sp_regime;
    }
    c->sp_last = false;
    if (!want) { c->sp_count = 0; return DPMM_OK; }
    c->sp_count = (K != c->sp_K) ? 1u : c->sp_count + 1u;      // (a new number of clusters: measure again)
    c->sp_K = K;
    HIPCHK(c, launch_niw_direction(c->d_Rp, c->d_mup, c->d_cst, c->D, K, c->d_sp_frag, c->d_sp_cons, c->stream));
    c->sp_ready = true;
    return DPMM_OK;
}

// compat entry points: copy the caller's cluster-ordered arrays into the staging rows (identity slot map) and commit
static int stage_and_commit(dpmm_ctx *c, int K, const float *mu, const float *mat, const float *logdet, const float *lr, const float *w) {
    float *smu, *smat, *sld, *slr, *sw;
    int32_t *sslot;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_stream(c, c->stream));          // an earlier pack kernel may still read the staging buffer
    if (int rc = dpmm_params_staging(c, K, &smu, &smat, &sld, &slr, &sw, &sslot)) return rc;
    const size_t D = (size_t)c->D;
    if (mu) memcpy(smu, mu, sizeof(float) * 3 * K * D);
    if (c->prior == DPMM_PRIOR_NIW) {       // full row-major R from the caller -> the packed upper triangle of the staging rows
        const size_t T = D * (D + 1) / 2;
        for (size_t j = 0; j < 3 * (size_t)K; ++j)
            for (size_t r = 0; r < D; ++r) memcpy(smat + j * T + r * D - r * (r - 1) / 2, mat + (j * D + r) * D + r, sizeof(float) * (D - r));
    } else {
        memcpy(smat, mat, sizeof(float) * 3 * K * D);
    }
    if (logdet) memcpy(sld, logdet, sizeof(float) * 3 * K);
    memcpy(slr, lr, sizeof(float) * 2 * K);
    memcpy(sw, w, sizeof(float) * K);
    for (int k = 0; k < K; ++k) sslot[k] = k;
    return dpmm_commit_params(c, K);
}

int dpmm_set_params_niw_chol(dpmm_ctx *c, int K, const float *mu, const float *R, const float *logdet, const float *lr, const float *w) {
    if (!c) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_NIW) return fail(c, DPMM_EINVAL, "context was created for another prior");
    if (!mu || !R || !logdet || !lr || !w) return fail(c, DPMM_EINVAL, "null parameter array");
    if (int rc = check_K(c, K)) return rc;
    return stage_and_commit(c, K, mu, R, logdet, lr, w);
}

int dpmm_set_params_niw(dpmm_ctx *c, int K, const float *mu, const float *inv_sigma, const float *logdet, const float *lr, const float *w) {
    if (!c) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_NIW) return fail(c, DPMM_EINVAL, "context was created for another prior");
    if (!inv_sigma) return fail(c, DPMM_EINVAL, "null parameter array");
    if (int rc = check_K(c, K)) return rc;
    // compat path: Sigma^-1 = R'R (upper Cholesky factor) in Float64 on the host
    const int D = c->D;
    std::vector<float> R((size_t)3 * K * D * D, 0.f);
    std::vector<double> L((size_t)D * D);
    for (int j = 0; j < 3 * K; ++j) {
        const float *A = inv_sigma + (size_t)j * D * D;
        std::fill(L.begin(), L.end(), 0.0);
        for (int a = 0; a < D; ++a) {  // lower factor, row by row
            for (int b = 0; b <= a; ++b) {
                double s = 0.5 * ((double)A[(size_t)a * D + b] + (double)A[(size_t)b * D + a]);
                for (int t = 0; t < b; ++t) s -= L[(size_t)a * D + t] * L[(size_t)b * D + t];
                L[(size_t)a * D + b] = (a == b) ? std::sqrt(s) : s / L[(size_t)b * D + b];
            }
        }
        float *Rj = R.data() + (size_t)j * D * D;
        for (int a = 0; a < D; ++a)
            for (int b = a; b < D; ++b) Rj[(size_t)a * D + b] = (float)L[(size_t)b * D + a];  // R = L'
    }
    return dpmm_set_params_niw_chol(c, K, mu, R.data(), logdet, lr, w);
}

int dpmm_set_params_mult(dpmm_ctx *c, int K, const float *logp, const float *lr, const float *w) {
    if (!c) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_MULT) return fail(c, DPMM_EINVAL, "context was created for another prior");
    if (!logp || !lr || !w) return fail(c, DPMM_EINVAL, "null parameter array");
    if (int rc = check_K(c, K)) return rc;
    return stage_and_commit(c, K, nullptr, logp, nullptr, lr, w);
}

int dpmm_set_num_clusters(dpmm_ctx *c, int K) {
    if (!c) return DPMM_EINVAL;
    if (int rc = check_K(c, K)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_capacity(c, K)) return rc;
    if (K != c->K) c->have_params = false;
    c->K = K;
    return DPMM_OK;
}

int dpmm_num_clusters(const dpmm_ctx *c) { return c ? c->K : 0; }

int dpmm_numa_node(dpmm_ctx *c) {
    if (!c) return -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), c->device) != hipSuccess) return -1;
    for (char *p = bus; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');      // sysfs uses lower-case hex
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

static int noise_flush(dpmm_ctx *c);
static int run_sweep(dpmm_ctx *c, uint32_t epoch, int final_argmax, float *table, int64_t table_stride) {
    if (!c->have_points || !c->have_params) return fail(c, DPMM_ESTATE, "sweep needs points and parameters");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) return DPMM_OK;
    if (c->prior == DPMM_PRIOR_NIW && !table) {
        if (!c->work_zeroed) HIPCHK(c, hipMemsetAsync(c->d_work, 0, sizeof(unsigned long long) * DPMM_WORK_SLOTS, c->stream));   // usually done by the pack kernel
        c->work_zeroed = false;
    }
    if (!table && (c->opt_timing & 1)) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (c->prior == DPMM_PRIOR_NIW) {
        NiwSweepArgs a{};
        a.X = c->dX; a.ldx = c->ldx; a.n = c->n; a.first_index = c->first; a.ntiles = c->ntiles; a.K = c->K;
        a.Rp = c->d_Rp; a.mup = c->d_mup; a.cst = c->d_cst;
        a.tdf = c->predictive ? c->d_tdf : nullptr;
        a.scratch = table ? table : c->d_scratch;
        a.scratch_stride = table ? table_stride : c->scratch_stride;
        a.scratch_by_tile = table ? 1 : 0;
        a.labels_only = table ? 1 : 0;
        a.bins = c->dbins; a.seed = c->seed; a.epoch = epoch; a.final_argmax = final_argmax;
        {
            const bool no_order = !c->opt_ordered;
            a.screen_margin = table ? 0.f : c->opt_margin;
            a.use_prev = c->have_labels ? 1 : 0;
            a.lam = (c->have_screen_prep && !c->predictive) ? c->d_lam : nullptr;
            a.mdist = c->d_mdist;
            a.tail = (c->have_tail && !c->predictive) ? c->d_tail : nullptr;
            a.tail_g = ((c->D - 4) % 16) / 4;
            a.ball = c->opt_ball;
            a.bracket = c->opt_bracket;
            a.bf16scr = (c->opt_bf16scr && c->NB == 4 && a.tail != nullptr) ? 1 : 0;
            if (c->sp_ready && a.bf16scr && !table && !final_argmax) { c->sp_last = a.lam == nullptr && a.tdf == nullptr && a.screen_margin > 0.f && c->K > 1 && c->K <= SP_MAXK; /* (exactly the conditions of launch_direct's DIR instantiation -- the kernel that writes the yield words; K <= 64 fits the LDS table) */ a.sp_frag = c->d_sp_frag; a.sp_cons = c->d_sp_cons; if (c->opt_direction > 0 || c->sp_count % 32u != 1u) a.bf16scr |= 2; }
            a.need = (table || c->opt_direction == 0) ? nullptr : c->h_need;
            // the visiting order is set BEFORE the bracket launch: its tiles and thresholds are indexed by visiting position and
            // must be those the sweep walks (a bracket computed in storage order would hand a point another point's threshold)
            a.order = (!table && c->have_perm && !no_order) ? c->sb.perm : nullptr;
            a.order_total = c->sb.perm_total;
            if (c->have_refb_big && a.tail != nullptr && !table && a.bracket && a.use_prev && a.screen_margin > 0.f && !final_argmax) {
                // D = 128, 256: the reference bracket as a launch of its own in front of the sweep (niw_bracket_big_kernel): per tile whether all
                // its points were in one cluster, per point the lower end of a certified bracket of that cluster's value
                if (!c->d_brk_flag) {
                    const size_t nt = (size_t)((c->n + 127) / 128);
                    HIPCHK(c, hipMalloc(&c->d_brk_flag, sizeof(uint32_t) * std::max<size_t>(nt, 1)));
                    HIPCHK(c, hipMalloc(&c->d_brk_aref, sizeof(float) * std::max<size_t>(nt * 128, 1)));
                }
                HIPCHK(c, launch_niw_bracket_big(c->NB, a, c->d_refb_big, c->d_brk_flag, c->d_brk_aref, c->stream));
                a.sp_frag = c->d_brk_flag; a.sp_cons = c->d_brk_aref;
            }
            a.work = table ? nullptr : c->d_work;
            if (!table) { c->work_waves = std::max(c->work_waves, 4 * c->sweep_grid); c->work_launches += 1; }
            a.prio = c->opt_prio;
            a.queue_rounds = c->opt_queue_rounds;
        }
#ifdef DPMM_STAMPS
        if (!g_dbg) { hipMalloc(&g_dbg, sizeof(unsigned long long) * 16 * 4 * 4096); hipMemset(g_dbg, 0, sizeof(unsigned long long) * 16 * 4 * 4096); }
        a.dbg = g_dbg;
#endif
        const bool b3 = c->NB == 4 && c->have_b3 && c->opt_b3 && !table && !c->predictive && a.tail != nullptr;
        if (b3) {
            // bf16 sub-label evaluation active (niw_lean.hip): the sweep kernel draws and STORES the labels (LSTORE instantiations), the sub-labels of
            // the same tiles follow in a launch of their own -- every sub-cluster value of a sweep then comes from the same arithmetic
            // First the tiles the cheap screens settle completely (niw_lean_kernel: labels and sub-labels in one go); what it leaves goes through
            // the list.  A sweep that left more than 30 % of its tiles (overlapping clusters: the screens' later stages do the work) switches the
            // lean kernel off for 15 sweeps -- every path gives the same labels and sub-labels, so the decision is free to be local.
            // (round 6: the lean kernel runs the direction screen itself while the tables exist -- it used to stay off in that regime)
            const bool lean_ok = c->opt_lean != 0 && c->opt_bracket && a.use_prev && a.screen_margin > 0.f && !final_argmax && a.lam == nullptr && c->K > 1 && c->d_hard &&
                                 !(c->opt_lean_dir == 0 && c->sp_regime && c->opt_direction != 0 && c->sp_ready);
            const int64_t nwt = (c->n + 63) / 64;
            if (lean_ok && c->lean_off == 0 && c->lean_ran) {            // the verdict on the last sweep that ran the lean kernel
                if ((int64_t)c->h_hard[0] * 10 > nwt * 3) { c->lean_off = c->lean_backoff; c->lean_backoff = std::min(2 * c->lean_backoff + 1, 1023); }
                else c->lean_backoff = 15;
                c->lean_ran = false;
            }
            const bool use_lean = lean_ok && c->lean_off == 0;
            if (lean_ok && c->lean_off > 0) { c->lean_off -= 1; c->h_hard[0] = 0; }
            const uint32_t *list = nullptr;
            const bool parts = (c->opt_timing & 8) != 0 && (c->opt_timing & 1) != 0;
            if (parts && !c->ev_part[0]) for (auto &e : c->ev_part) HIPCHK(c, hipEventCreate(&e));
            if (use_lean) {
                const size_t hw = hard_list_words(c->n);
                uint32_t *mine = c->d_hard + (size_t)c->hard_flip * hw, *other = c->d_hard + (size_t)(c->hard_flip ^ 1) * hw;
                c->hard_flip ^= 1;
                c->lean_ran = true;
                uint32_t *need2 = a.need ? c->h_need + 8 * (size_t)c->sweep_grid_max : nullptr;
                // (tiles aligned to the bins of the sort that wrote the visiting order, when there is one and it fits the kernel's table)
                const bool al = a.order != nullptr && c->perm_nbins > 0 && c->perm_nbins <= NIW_LEAN_MAX_BINS;
                uint32_t *need3 = a.need ? c->h_need + 12 * (size_t)c->sweep_grid_max : nullptr;
                NiwSweepArgs al_args = a;
                if (!c->opt_lean_dir) { al_args.sp_frag = nullptr; al_args.sp_cons = nullptr; }
                if (c->have_pb && c->opt_pair_ball && al_args.ball) al_args.ball |= 2;      // (a bit of `ball`: a field of its own re-lays every kernel's scalars)
                HIPCHK(c, launch_niw_lean(al_args, mine, need2, other, al ? c->sb.bin_start : nullptr, al ? c->perm_nbins : 0, need3, c->sweep_grid, c->stream));
                if (parts) HIPCHK(c, hipEventRecord(c->ev_part[0], c->stream));
                list = mine;
            }
            a.bf16scr |= 4;
            a.tdf = reinterpret_cast<const float *>(list);                         // (the LSTORE instantiations' tile list: null = all tiles)
            const float *mdist_kept = a.mdist;
            if (list) a.mdist = reinterpret_cast<const float *>(c->h_hard);        // (LIST instantiations: where the list's length goes; they also finish the spans' sub-labels)
            HIPCHK(c, launch_niw_sweep(c->NB, a, c->sweep_grid, c->stream));
            a.mdist = mdist_kept;
            if (parts) HIPCHK(c, hipEventRecord(c->ev_part[1], c->stream));
            c->have_parts = parts ? (use_lean ? 2 : 1) : 0;
            a.bf16scr &= 3;
            a.tdf = nullptr;
            if (!list) HIPCHK(c, launch_niw_sub(a, nullptr, nullptr, c->sweep_grid, c->stream));      // (every tile; a list's spans were finished by the LIST launch)
        } else {
            HIPCHK(c, launch_niw_sweep(c->NB, a, c->sweep_grid, c->stream));
            if (!table) c->have_parts = 0;          // (a one-launch sweep records no part events: dpmm_last_sweep_parts_ms must not mix its ev[0] / ev[1] with an older sweep's)
        }
    } else {
        MultSweepArgs a{};
        a.X = c->dX; a.ldx = c->ldx; a.n = c->n; a.first_index = c->first; a.D = c->D; a.K = c->K;
        a.logp = c->d_Rp; a.cst = c->d_cst;
        a.scratch = table ? table : c->d_scratch;
        a.scratch_stride = table ? table_stride : c->scratch_stride;
        a.scratch_by_tile = table ? 1 : 0;
        a.labels_only = table ? 1 : 0;
        a.bins = c->dbins; a.seed = c->seed; a.epoch = epoch; a.final_argmax = final_argmax;
        a.use_prev = c->have_labels ? 1 : 0;
        a.order = (!table && c->have_perm && c->opt_ordered) ? c->sb.perm : nullptr;
        a.order_total = c->sb.perm_total;
        if (c->x_u8) HIPCHK(c, launch_mult_sweep_u8(a, c->dX8, c->ld8, c->d_Lp16, c->sweep_grid, c->stream));
        else if (c->x_bf16_exact) HIPCHK(c, launch_mult_sweep_bf16(a, c->d_Lp16, c->sweep_grid, c->stream));
        else {
            if (!c->rp_current) { HIPCHK(c, launch_mult_pack(c->d_raw, c->d_Rp, 3 * c->K, c->ldx, c->stream)); c->rp_current = true; }      // (skipped by a device-master draw)
            HIPCHK(c, launch_mult_sweep(a, c->sweep_grid, c->stream));
        }
    }
    if (!table) {
        if (c->master) if (int rc = noise_flush(c)) return rc;
        if (c->opt_timing & 1) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        c->have_sweep_ev = (c->opt_timing & 1) != 0;
        c->have_labels = true;
    }
    return DPMM_OK;
}

int dpmm_sweep(dpmm_ctx *c, uint32_t epoch, int final_argmax) {
    if (!c) return DPMM_EINVAL;
    if (c->predictive) return fail(c, DPMM_ESTATE, "predictive parameters are loaded: set the sweep parameters again");
    return run_sweep(c, epoch, final_argmax, nullptr, 0);
}

int dpmm_set_predictive_niw(dpmm_ctx *c, int K, const float *m, const float *R, const float *logdet, const float *df, const float *w) {
    if (!c) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_NIW) return fail(c, DPMM_EINVAL, "context was created for another prior");
    if (!m || !R || !logdet || !df || !w) return fail(c, DPMM_EINVAL, "null parameter array");
    if (int rc = check_K(c, K)) return rc;
    const size_t D = (size_t)c->D;
    // expand to the (cluster, left, right) row layout of the sweep kernels; only rows 3k are read in table mode
    std::vector<float> mu3(3 * K * D, 0.f), R3(3 * K * D * D, 0.f), ld3(3 * (size_t)K, 0.f), lr(2 * (size_t)K, 0.5f);
    for (int k = 0; k < K; ++k) {
        memcpy(&mu3[(size_t)3 * k * D], m + (size_t)k * D, sizeof(float) * D);
        memcpy(&R3[(size_t)3 * k * D * D], R + (size_t)k * D * D, sizeof(float) * D * D);
    }
    if (int rc = dpmm_set_params_niw_chol(c, K, mu3.data(), R3.data(), ld3.data(), lr.data(), w)) return rc;
    // MvTDist log-density constant: lgamma((df+D)/2) - lgamma(df/2) - D/2 log(df pi) - logdet(Sigma)/2 + log w
    std::vector<float> cst(3 * (size_t)K, 0.f), tdf(6 * (size_t)K, 1.f);
    for (int k = 0; k < K; ++k) {
        const double v = df[k];
        cst[3 * k] = (float)(lgamma(0.5 * (v + D)) - lgamma(0.5 * v) - 0.5 * D * log(v * M_PI) - 0.5 * logdet[k] + log((double)w[k]));
        tdf[6 * k] = (float)v;
        tdf[6 * k + 1] = (float)(0.5 * (v + D));
    }
    if (int rc = ensure_pinned(c, sizeof(float) * 9 * K)) return rc;
    HIPCHK(c, sync_stream(c, c->stream));
    memcpy(c->h_pin, cst.data(), sizeof(float) * 3 * K);
    memcpy(c->h_pin + sizeof(float) * 3 * K, tdf.data(), sizeof(float) * 6 * K);
    HIPCHK(c, launch_copy_bytes(c->d_cst, c->h_pin, sizeof(float) * 3 * K, c->stream));
    HIPCHK(c, launch_copy_bytes(c->d_tdf, c->h_pin + sizeof(float) * 3 * K, sizeof(float) * 6 * K, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    c->predictive = true;
    return DPMM_OK;
}

int dpmm_set_predictive_mult(dpmm_ctx *c, int K, const float *logp, const float *w) {
    if (!c) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_MULT) return fail(c, DPMM_EINVAL, "context was created for another prior");
    if (!logp || !w) return fail(c, DPMM_EINVAL, "null parameter array");
    if (int rc = check_K(c, K)) return rc;
    const size_t D = (size_t)c->D;
    std::vector<float> lp3(3 * K * D, 0.f), lr(2 * (size_t)K, 0.5f);
    for (int k = 0; k < K; ++k) memcpy(&lp3[(size_t)3 * k * D], logp + (size_t)k * D, sizeof(float) * D);
    if (int rc = dpmm_set_params_mult(c, K, lp3.data(), lr.data(), w)) return rc;
    c->predictive = true;
    return DPMM_OK;
}

int dpmm_predict(dpmm_ctx *c, float *parr) {
    if (!c || !parr) return DPMM_EINVAL;
    if (!c->predictive) return fail(c, DPMM_ESTATE, "dpmm_predict needs dpmm_set_predictive_* first");
    return dpmm_debug_loglik(c, parr);
}

int dpmm_predict_points(dpmm_ctx *c, int64_t *labels, float *probs) {
    if (!c || !labels) return DPMM_EINVAL;
    if (!c->predictive) return fail(c, DPMM_ESTATE, "dpmm_predict_points needs dpmm_set_predictive_* first");
    if (!c->have_points || !c->have_params) return fail(c, DPMM_ESTATE, "predict needs points and parameters");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) return DPMM_OK;
    const int64_t stride = c->ntiles * c->tile;
    const int rstep = (c->prior == DPMM_PRIOR_NIW) ? 1 : 3;       // Multinomial: rows 3k are the cluster-level rows
    float *table = nullptr, *d_probs = nullptr;
    int64_t *d_lab = nullptr;
    HIPCHK(c, hipMalloc(&table, sizeof(float) * (size_t)(rstep * c->K) * (size_t)stride));
    int rc = run_sweep(c, 0, 0, table, stride);
    hipError_t e = hipSuccess;
    if (rc == DPMM_OK) {
        e = hipMalloc(&d_lab, sizeof(int64_t) * (size_t)c->n);
        if (e == hipSuccess && probs) e = hipMalloc(&d_probs, sizeof(float) * (size_t)c->n * (size_t)c->K);
        if (e == hipSuccess) e = launch_predict_finish(table, stride, rstep, c->n, c->K, d_lab, d_probs, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(labels, d_lab, sizeof(int64_t) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && probs) e = hipMemcpyAsync(probs, d_probs, sizeof(float) * (size_t)c->n * (size_t)c->K, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = sync_stream(c, c->stream);
        if (e != hipSuccess) { c->err = std::string("dpmm_predict_points: ") + hipGetErrorString(e); rc = DPMM_EHIP; }
    }
    hipFree(table); hipFree(d_lab); hipFree(d_probs);
    return rc;
}

int dpmm_debug_loglik(dpmm_ctx *c, float *out) {
    if (!c || !out) return DPMM_EINVAL;
    if (!c->have_points || !c->have_params) return fail(c, DPMM_ESTATE, "debug_loglik needs points and parameters");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) return DPMM_OK;
    const int64_t stride = c->ntiles * c->tile;
    const int rows = (c->prior == DPMM_PRIOR_NIW) ? c->K : 3 * c->K;
    float *table = nullptr;
    HIPCHK(c, hipMalloc(&table, sizeof(float) * (size_t)rows * (size_t)stride));
    int rc = run_sweep(c, 0, 0, table, stride);
    if (rc == DPMM_OK) {
        // Multinomial: rows 3k of the 3K-row table are the cluster-level rows
        const size_t src_pitch = sizeof(float) * stride * (c->prior == DPMM_PRIOR_MULT ? 3 : 1);
        hipError_t e = hipMemcpy2DAsync(out, sizeof(float) * c->n, table, src_pitch, sizeof(float) * c->n,
                                        (size_t)c->K, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = sync_stream(c, c->stream);
        if (e != hipSuccess) { c->err = std::string("dpmm_debug_loglik: ") + hipGetErrorString(e); rc = DPMM_EHIP; }
        else if (c->opt_ref_const && c->prior == DPMM_PRIOR_NIW && !c->predictive) {
            // mv_gaussian.jl:24 normalises with length(Sigma) = D^2: -(D^2 log 2 pi + logdet)/2; the table carries -logdet/2 only
            const float k2pi = (float)(-0.5 * (double)c->D * (double)c->D * log(2.0 * M_PI));
            for (size_t i = 0; i < (size_t)c->K * (size_t)c->n; ++i) out[i] += k2pi;
        }
    }
    hipFree(table);
    return rc;
}

int64_t dpmm_packed_stride(const dpmm_ctx *c) { return c ? c->packed_stride : 0; }

static int ensure_out(dpmm_ctx *c, size_t bytes) {
    if (bytes <= c->h_out_bytes) return DPMM_OK;
    HIPCHK(c, sync_stream(c, c->stream));
    c->rows_late = false;
    if (c->h_out) hipHostFree(c->h_out);
    c->h_out = nullptr; c->h_out_bytes = 0;
    size_t cap = 1 << 20;
    while (cap < bytes) cap *= 2;
    HIPCHK(c, hipHostMalloc((void **)&c->h_out, cap, hipHostMallocDefault));
    c->h_out_bytes = cap;
    return DPMM_OK;
}

// One-collective pass: the sub-labels of this shard's candidates that turned out not to be bad go back to the side they were on.  Labels
// only -- nothing the master waits for -- so the kernel is queued behind the posteriors / the copy of the rows, before the call returns
// (every later reader of the labels is stream-ordered behind it).
static int flush_undo(dpmm_ctx *c) {
    if (!c->undo_pending) return DPMM_OK;
    c->undo_pending = false;
    const uint8_t *flags = reinterpret_cast<const uint8_t *>(c->d_out) + sizeof(double) * 2 * (size_t)c->K * (size_t)c->packed_stride;
    HIPCHK(c, launch_niw_undo_reset(c->dbins, c->n, c->K, flags, c->d_cside, c->stream));
    return DPMM_OK;
}
// One statistics pass on the ctx stream (asynchronous): [sub-cluster occupancies -> bad-cluster reset ->] sort by bin ->
// segmented statistics -> packed rows in c->d_out [-> all-reduce over the ranks].
// with_reset (the per-step pass) has two forms when a communicator is attached:
//   two collectives (DPMM_OPT_ONE_COLLECTIVE = 0, Multinomial): histogram -> ALL-REDUCE of the 2K Int64 occupancies -> flags + reset +
//       re-count -> scan / scatter -> statistics -> ALL-REDUCE of the 2K packed rows.  The first collective sits in the middle of the sort chain.
//   one collective (NIW, default): the reset is applied SPECULATIVELY to this shard's candidates -- clusters with exactly one empty
//       sub-cluster HERE; a cluster that is bad globally is a candidate on every rank that holds points of it -- and ONE ALL-REDUCE
//       carries 3K rows: the 2K rows of the labels AS SWEPT (for a candidate: everything on the side its points were on, rebuilt from
//       left' + right') and K re-drawn left rows X.  Behind it niw_finalize_rows_kernel reads the verdict off the rows' own N column:
//       bad cluster -> (X, left + right - X), anything else -> the rows as they are; a shard's candidate that is not bad gets its
//       sub-labels back (all of its points were on one side).  No history, no second pass, 1.5x the bytes of a 1.1 MB message.
//   Same end state as the reference's full pass + subset pass (update_suff_stats_posterior! + reset_bad_clusters!,
//   src/local_clusters_actions.jl:665-666, 501-516) either way; tests/test_gpu_multirank.py: the chains are the one-rank chain.
static int run_stats(dpmm_ctx *c, const int64_t *idx, int n_idx, bool with_reset = false, uint32_t reset_epoch = 0, uint8_t *flags_to = nullptr,
                     bool *flags_sent = nullptr) {
    if (!c->have_points || !c->have_labels || c->K < 1) return fail(c, DPMM_ESTATE, "suffstats need points, labels and parameters (K)");
    HIPCHK(c, hipSetDevice(c->device));
    const int nbins = 2 * c->K;
    c->marg_valid = false;
    c->mspec_valid = false;            // (draws made ahead belong to the rows this pass replaces)
    // automatic: the one-collective form sends 3K rows instead of 2K -- 0.55 MB more at D = 64, K = 32 (a few us of wire) against a whole
    // latency-bound collective in the middle of the sort chain; at D = 256 the extra K rows are 8.4 MB (~80 us): the classic form stays
    const bool one_auto = c->packed_stride <= 4096;
    const bool one_coll = with_reset && comm_attached(c) && (c->opt_one_collective < 0 ? one_auto : c->opt_one_collective != 0);      // (both priors since round 5)
    c->last_pass_one_collective = one_coll;
    if (c->opt_timing & 2) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    if (idx) {
        c->h_sel.assign(nbins, 0);
        for (int j = 0; j < n_idx; ++j) {
            if (idx[j] < 1 || idx[j] > c->K) return fail(c, DPMM_EINVAL, "cluster index out of range");
            c->h_sel[2 * (idx[j] - 1)] = 1;
            c->h_sel[2 * (idx[j] - 1) + 1] = 1;
        }
        if (int rc = ensure_pinned(c, nbins + 8)) return rc;
        HIPCHK(c, sync_stream(c, c->stream));
        memcpy(c->h_pin, c->h_sel.data(), nbins);
        HIPCHK(c, launch_copy_bytes(c->sb.bin_sel, c->h_pin, nbins, c->stream));
        c->sel_all_ones = 0;
    } else if (with_reset && c->n > 0 && c->opt_derive) {
        // (the per-step pass writes the selection of its 2K bins itself: starts_step_kernel)
    } else if (c->sel_all_ones < nbins) {
        HIPCHK(c, hipMemsetAsync(c->sb.bin_sel, 1, c->sel_capacity, c->stream));      // stays valid until a subset pass overwrites it
        c->sel_all_ones = c->sel_capacity;
    }
    bool derive = false;
    StatsArgs a{};
    a.X = c->dX; a.ldx = c->ldx; a.n = c->n; a.D = c->D; a.nbins = nbins; a.chunk = c->chunk;
    a.max_items = (int)((c->n + c->chunk - 1) / c->chunk) + nbins;
    a.range_groups = c->opt_stats_groups;
    a.sb = c->sb; a.slabs = c->d_slabs; a.slab_stride = c->slab_stride; a.out = (one_coll && c->prior == DPMM_PRIOR_NIW) ? c->d_red : c->d_out; a.packed_stride = c->packed_stride; a.row_off = c->d_row_off; a.inv_off = c->d_inv_off;      // (a.out, NIW one-collective pass: the reduce kernel writes the 3K travelling rows itself)
    if (with_reset && c->n > 0) {
        // reset_bad_clusters! (local_clusters_actions.jl:501-516) on the device, four launches: histogram (+ running bin totals) ->
        // [occupancies summed over the ranks] -> flags + sub-labels of flagged clusters re-drawn + touched tiles re-counted -> scan + starts
        // -> scatter.  The flags live right behind the packed rows, so that rows + flags reach the master in one copy.
        // DPMM_OPT_CHAIN_FUSION bit 8 (round 6): no reset launch -- the histogram counts the reset ahead for the clusters that are one-sided in a
        // tile, the scan derives the flags and picks those counts for flagged clusters, the scatter applies the re-draw while it places
        // (measured, round 6: the launch it removes costs 18 us at N = 1e7 and 7 us at the 8-GPU shard size; the three kernels that take its work
        //  over grow by 6.5 + 4.3 + 9.2 us and 2.4 + 2.0 + 3.4 us -- no gain at either size, so the bit is OFF by default; it is value-neutral and
        //  tested (tests/test_gpu_derive.py).  Bit 8 alone applies from 4e6 points per shard, bit 32 at any size.)
        const bool fold_reset = (c->opt_chain & 8) != 0 && (c->n >= 4000000 || (c->opt_chain & 32) != 0) && nbins <= STEP_SPEC_MAX_BINS &&
                                c->sb.tile_spec != nullptr && c->sb.spec_bins != nullptr;
        HIPCHK(c, launch_step_hist(c->dbins, c->n, nbins, c->sb, c->stream, fold_reset ? 1 : 0, c->first, c->seed, reset_epoch));
        const long long *gc = nullptr;
        if (comm_attached(c) && !one_coll) {
            HIPCHK(c, launch_widen_counts(c->sb.fast_total, FAST_TOTAL_STRIDE, c->d_counts64, nbins, c->stream));
            if (int rc = comm_allreduce(c, c->d_counts64, nbins, /*kind=*/0)) return rc;
            gc = c->d_counts64;
        }
        uint8_t *flags = reinterpret_cast<uint8_t *>(c->d_out) + sizeof(double) * (size_t)nbins * (size_t)c->packed_stride;
        // (one collective: this shard's own occupancies decide which clusters are reset speculatively; the verdict follows the all-reduce)
        if (!fold_reset) HIPCHK(c, launch_step_reset(c->dbins, c->n, c->first, nbins, c->sb, gc, flags, c->K, c->seed, reset_epoch, one_coll ? c->d_cside : nullptr, c->stream));
        // Statistics of the SMALLER sub-cluster only wherever no point entered or left the cluster since its cluster-level row was
        // cached (labels are tracked by the histogram); the other sub-cluster is cache - computed (derive_rows_kernel below)
        derive = c->opt_derive != 0;
        const int force_all = (c->cache_force || c->cache_K != c->K) ? 1 : 0;
        const StepReset rs{gc, flags, c->K, one_coll ? c->d_cside : nullptr, c->first, c->seed, reset_epoch};
        HIPCHK(c, launch_step_scan_scatter(c->dbins, a, derive ? 1 : 0, force_all, c->opt_chain & 1, fold_reset ? &rs : nullptr, c->stream));
        if (derive) { c->sel_all_ones = 0; c->cache_force = false; c->cache_K = c->K; }
    } else {
        if (c->n > 0) {
            HIPCHK(c, launch_sort_by_bin(c->dbins, c->n, nbins, c->sb, c->stream));
        } else {
            HIPCHK(c, hipMemsetAsync(c->sb.bin_total, 0, sizeof(int32_t) * nbins, c->stream));
        }
        if (with_reset && one_coll) {
            HIPCHK(c, hipMemsetAsync(c->d_cside, 0, (size_t)c->K, c->stream));      // a rank without points has no candidates and adds zeros
        } else if (with_reset) {       // a rank without points: its occupancies are zero, the flags come from the other ranks' counts
            const long long *gc = nullptr;
            if (comm_attached(c)) {
                HIPCHK(c, launch_widen_counts(c->sb.bin_total, 1, c->d_counts64, nbins, c->stream));
                if (int rc = comm_allreduce(c, c->d_counts64, nbins, /*kind=*/0)) return rc;
                gc = c->d_counts64;
            }
            uint8_t *flags = reinterpret_cast<uint8_t *>(c->d_out) + sizeof(double) * (size_t)nbins * (size_t)c->packed_stride;
            HIPCHK(c, launch_bad_flags(c->sb.bin_total, gc, c->K, flags, c->stream));
        }
        HIPCHK(c, launch_sort_finish(c->dbins, a, c->stream));
    }
    c->have_perm = c->n > 0;
    c->perm_nbins = nbins;
    if (flags_sent) *flags_sent = false;
    if (one_coll) {
        uint8_t *flags = reinterpret_cast<uint8_t *>(c->d_out) + sizeof(double) * (size_t)nbins * (size_t)c->packed_stride;
        const size_t nred = 3 * (size_t)c->K * (size_t)c->packed_stride;
        if (c->n > 0 && c->prior == DPMM_PRIOR_NIW) {
            if (derive) { a.mode = c->sb.cmode; a.cache = c->d_ccache; a.dirty = c->sb.cdirty; a.K = c->K; }
            a.cside = c->d_cside; a.zero2 = flags + c->K;
            HIPCHK(c, launch_niw_stats(a, c->stream));
        } else if (c->n > 0) {
            // Multinomial: the usual statistics + derivation of the speculatively reset labels into d_out, then the travelling rows from them
            if (c->x_u8) HIPCHK(c, launch_mult_stats_u8(a, c->dX8, c->ld8, c->stream));
            else HIPCHK(c, launch_mult_stats(a, c->stream));
            if (derive) HIPCHK(c, launch_derive_rows(c->d_out, c->d_ccache, c->sb.cmode, c->sb.cdirty, c->packed_stride, c->K, flags, nullptr, c->stream));
            HIPCHK(c, launch_onecoll_rows(c->d_out, c->d_red, c->packed_stride, c->K, c->d_cside, flags, c->stream));
        } else {
            HIPCHK(c, hipMemsetAsync(c->d_red, 0, sizeof(double) * nred, c->stream));
            HIPCHK(c, hipMemsetAsync(flags + c->K, 0, 2, c->stream));
        }
        // the one exchange of the sweep (update_suff_stats_posterior!, local_clusters_actions.jl:206-254; aggregate_suff_stats)
        if (int rc = comm_allreduce(c, c->d_red, nred, /*kind=*/1)) return rc;
        // (flags_to: the caller's pinned block -- the verdict rides in this launch; its "any" byte starts at 0: the kernel only sets it, and
        // nothing of the stream reads or writes the block before this kernel)
        if (flags_to) flags_to[c->K] = 0;
        HIPCHK(c, launch_niw_finalize_rows(c->d_red, c->d_out, c->packed_stride, c->K, c->d_cside, flags, flags_to, c->stream));
        if (flags_to && flags_sent) *flags_sent = true;
        // the undo of the candidates that are not bad touches labels only: queued by the caller BEHIND what the master is waiting for
        c->undo_pending = true;
        c->comm_bytes[0] = 0;
        if (c->opt_timing & 2) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
        c->have_stats_ev = (c->opt_timing & 2) != 0;
        c->rows_full_K = c->K;
        return DPMM_OK;
    }
    // (flags_to: the caller's pinned block for the bad-cluster flags -- they ride in the derivation's launch when no collective follows it)
    const bool ride = derive && flags_to != nullptr && !comm_attached(c);
    const uint8_t *fsrc = reinterpret_cast<const uint8_t *>(c->d_out) + sizeof(double) * (size_t)nbins * (size_t)c->packed_stride;
    if (c->prior == DPMM_PRIOR_NIW) {
        if (derive) {       // the reduce kernel derives the rows that were not computed (one launch and one round trip of the rows less)
            a.mode = c->sb.cmode; a.cache = c->d_ccache; a.dirty = c->sb.cdirty; a.K = c->K;
            a.flags_src = ride ? fsrc : nullptr; a.flags_dst = ride ? flags_to : nullptr;
        }
        HIPCHK(c, launch_niw_stats(a, c->stream));
    } else {
        if (c->x_u8) HIPCHK(c, launch_mult_stats_u8(a, c->dX8, c->ld8, c->stream));
        else HIPCHK(c, launch_mult_stats(a, c->stream));
        if (derive) HIPCHK(c, launch_derive_rows(c->d_out, c->d_ccache, c->sb.cmode, c->sb.cdirty, c->packed_stride, c->K, fsrc, ride ? flags_to : nullptr, c->stream));
    }
    if (ride && flags_sent) *flags_sent = true;
    if (comm_attached(c)) {
        // the one exchange of the sweep: elementwise sum of the per-worker statistics (update_suff_stats_posterior!,
        // local_clusters_actions.jl:206-254; aggregate_suff_stats); N counts travel as Float64 integers (exact below 2^53)
        if (int rc = comm_allreduce(c, c->d_out, (size_t)nbins * (size_t)c->packed_stride, /*kind=*/1)) return rc;
    }
    if (c->opt_timing & 2) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    c->have_stats_ev = (c->opt_timing & 2) != 0;
    c->rows_full_K = idx ? -1 : c->K;           // (a subset pass zeroes the rows it was not asked for)
    return DPMM_OK;
}

int dpmm_bin_counts(dpmm_ctx *c, int64_t *counts) {
    if (!c || !counts) return DPMM_EINVAL;
    if (!c->have_points || !c->have_labels || c->K < 1) return fail(c, DPMM_ESTATE, "bin counts need points, labels and parameters (K)");
    HIPCHK(c, hipSetDevice(c->device));
    const int nbins = 2 * c->K;
    if (c->n == 0) { for (int b = 0; b < nbins; ++b) counts[b] = 0; return DPMM_OK; }
    if (int rc = ensure_pinned(c, sizeof(int32_t) * nbins)) return rc;
    HIPCHK(c, launch_sort_by_bin(c->dbins, c->n, nbins, c->sb, c->stream));      // hist + scan -> bin_total
    HIPCHK(c, launch_copy_bytes(c->h_pin, c->sb.bin_total, sizeof(int32_t) * nbins, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    const int32_t *h = reinterpret_cast<const int32_t *>(c->h_pin);
    for (int b = 0; b < nbins; ++b) counts[b] = h[b];
    return DPMM_OK;
}

int dpmm_suffstats_packed_device(dpmm_ctx *c, const int64_t *idx, int n_idx, double *d_out) {
    if (!c || !d_out) return DPMM_EINVAL;
    if (int rc = run_stats(c, idx, n_idx)) return rc;
    HIPCHK(c, hipMemcpyAsync(d_out, c->d_out, sizeof(double) * 2 * c->K * (size_t)c->packed_stride, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    return DPMM_OK;
}

// statistics (summed over the ranks when a communicator is attached) -> pinned host memory the caller reads in place
int dpmm_suffstats_host(dpmm_ctx *c, const int64_t *idx, int n_idx, const double **packed) {
    if (!c || !packed) return DPMM_EINVAL;
    const size_t out_bytes = sizeof(double) * 2 * (size_t)std::max(c->K, 1) * (size_t)c->packed_stride;
    if (int rc = ensure_out(c, out_bytes + DPMM_MAX_CLUSTERS + 64)) return rc;
    if (int rc = run_stats(c, idx, n_idx)) return rc;
    HIPCHK(c, launch_copy_bytes(c->h_out, c->d_out, out_bytes, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    *packed = reinterpret_cast<const double *>(c->h_out);
    return DPMM_OK;
}

static int mult_marginals_launch(dpmm_ctx *c);
int dpmm_step_stats(dpmm_ctx *c, uint32_t reset_epoch, const double **packed, const uint8_t **bad) {
    if (!c || !packed || !bad) return DPMM_EINVAL;
    const size_t out_bytes = sizeof(double) * 2 * (size_t)std::max(c->K, 1) * (size_t)c->packed_stride;
    if (int rc = ensure_out(c, out_bytes + DPMM_MAX_CLUSTERS + 64)) return rc;
    // (a step that will deliver the rows late lets the flags ride to the pinned block with the kernel that derives the rows: one copy fewer)
    const bool maybe_late = c->rows_on_demand && c->marg_req && c->opt_mult_draws_ahead && c->mdraw_seen && c->draws_on_device && c->d_raw2;
    bool flags_sent = false;
    if (int rc = run_stats(c, nullptr, 0, true, reset_epoch, maybe_late ? reinterpret_cast<uint8_t *>(c->h_out + out_bytes) : nullptr, &flags_sent)) return rc;
    if (c->marg_req) if (int rc = mult_marginals_launch(c)) return rc;       // the master's log-marginals ride behind the statistics: one wait
    const bool with_marg = c->marg_valid && c->marg_K == c->K;            // the device master is running (the host asked for the log-marginals)
    const bool ahead = with_marg && c->opt_mult_draws_ahead && c->mdraw_seen && c->draws_on_device && c->rows_full_K == c->K && c->d_raw2 &&
                       (!c->marg_req_outlier || c->mult_has_alpha1);
    const bool late = ahead && c->rows_on_demand;       // the caller reads the rows through dpmm_mult_master_rows_wait only: flags now, rows behind the draws
    c->rows_late = false;
    if (late) { if (!flags_sent) HIPCHK(c, launch_copy_bytes(c->h_out + out_bytes, reinterpret_cast<const uint8_t *>(c->d_out) + out_bytes, (size_t)c->K + 1, c->stream)); }
    else HIPCHK(c, launch_copy_bytes(c->h_out, c->d_out, out_bytes + (size_t)c->K + 1, c->stream));      // rows | flags
    if (int rc = flush_undo(c)) return rc;
    // Multinomial device master: the next Dirichlet draws + their hand-over images go out NOW, behind an event the host waits for instead of
    // the stream -- they run while it decides splits and merges (a quiet step then only uploads the weights; 24 us of kernels off the path)
    if (ahead) {
        if (!c->ev_rows) HIPCHK(c, hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_rows, c->stream));
        const int K = c->K;
        const uint32_t epoch = c->mdraw_epoch + 1;
        HIPCHK(c, launch_mult_dirichlet(c->d_out, c->packed_stride, c->d_malpha, c->mult_has_alpha1 ? c->d_malpha + c->ldx : nullptr, c->marg_req_outlier, c->D, c->ldx,
                                        K, c->seed, epoch, c->d_raw2, c->stream));
        c->rp2_current = !(c->x_u8 || c->x_bf16_exact);                 // (only the Float32 sweep kernel reads that image)
        if (c->rp2_current) HIPCHK(c, launch_mult_pack(c->d_raw2, c->d_Rp2, 3 * K, c->ldx, c->stream));
        if (c->x_u8) HIPCHK(c, launch_mult_pack_u8(c->d_raw2, c->d_Lp16_2, 3 * K, c->ldx, c->ld8, c->stream));
        else if (c->x_bf16_exact) HIPCHK(c, launch_mult_pack_bf16(c->d_raw2, c->d_Lp16_2, 3 * K, c->ldx, c->stream));
        c->mspec_valid = true; c->mspec_epoch = epoch; c->mspec_K = K; c->mspec_outlier = c->marg_req_outlier;
        if (late) {
            // (on the same stream, behind the draws.  A second stream was measured twice and is no faster: beside the Dirichlet kernel the copy
            //  slows that kernel down by what the copy takes, behind the draws it shares the PCIe link with the upload of the next weights.
            //  No event of its own either: dpmm_mult_master_rows_wait is rare and waits for the stream)
            HIPCHK(c, launch_copy_bytes(c->h_out, c->d_out, out_bytes, c->stream));
            c->rows_late = true;
        }
        HIPCHK(c, sync_event(c, c->ev_rows));
        c->sync_gen += 1;                  // (the weights' copy of this step was queued long before ev_rows)
        c->marg_behind_ev = true;
    } else {
        HIPCHK(c, sync_stream(c, c->stream));
    }
    *packed = reinterpret_cast<const double *>(c->h_out);
    *bad = reinterpret_cast<const uint8_t *>(c->h_out + out_bytes);
    return DPMM_OK;
}

// ---- the master's dense maths on the device --------------------------------------------------------------------------------
static int master_pinned(dpmm_ctx *c, size_t bytes) {
    if (bytes <= c->h_master_bytes) return DPMM_OK;
    HIPCHK(c, sync_stream(c, c->stream));
    if (c->h_master) hipHostFree(c->h_master);
    c->h_master = nullptr; c->h_master_bytes = 0;
    size_t cap = 1 << 16;
    while (cap < bytes) cap *= 2;
    HIPCHK(c, hipHostMalloc((void **)&c->h_master, cap, hipHostMallocDefault));
    c->h_master_bytes = cap;
    return DPMM_OK;
}
// the main stream waits for draws launched ahead on stream2 (before anything that writes what they read or reads what they write)
static int spec_join(dpmm_ctx *c) {
    if (c->spec_inflight) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_spec, 0));
        c->spec_inflight = false;
    }
    if (c->apairs_inflight) {             // (the pair job reads the stored rows a posterior pass is about to replace)
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_pairs, 0));
        c->apairs_inflight = false;
    }
    return DPMM_OK;
}
// ... and for the normals generated ahead (before a draw on the main stream: it uses them, or writes the buffer they are written to)
static int noise_join(dpmm_ctx *c) {
    if (c->noise_inflight) {
        // (a hipEventQuery in place of this barrier packet was tried: right after the record it reported the event's PREVIOUS completion
        // now and then -- draws from unwritten normals, caught by the moment test)
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_noise, 0));
        c->noise_inflight = false;
    }
    return DPMM_OK;
}
// the deferred launch of the normals of the next draws (requested by dpmm_niw_master_draw): into the draw buffer that is NOT in use
static int noise_flush(dpmm_ctx *c) {
    if (!c->noise_pending) return DPMM_OK;
    c->noise_pending = false;
    HIPCHK(c, launch_niw_master_noise(c->ma, c->noise_pend_nmat, c->noise_pend_epoch, c->d_Y[c->draw_cur ^ 1], c->stream2));
    HIPCHK(c, hipEventRecord(c->ev_noise, c->stream2));
    c->noise_inflight = true; c->noise_valid = true; c->noise_epoch = c->noise_pend_epoch; c->noise_nmat = c->noise_pend_nmat; c->noise_buf = c->draw_cur ^ 1;
    return DPMM_OK;
}
// the normals of the draws of `epoch` for K clusters are in d_Y[buf] (generated ahead on stream2)
static bool noise_ready(const dpmm_ctx *c, uint32_t epoch, int K, int buf) {
    return c->noise_valid && c->noise_epoch == epoch && 3 * K <= c->noise_nmat && c->noise_buf == buf;
}
// `data` [n] -> the device list `dst` (stream-ordered on the main stream; nothing is sent when the list is the one already there)
// The copy goes through a ring of pinned staging buffers and the library's own copy kernel, like every per-step transfer: on the test
// boxes hipMemcpyAsync takes the copy-engine path, which was seen to sit in a stream for 0.7 ms (rocprofv3: `__amd_rocclr_copyBuffer`).
// A staging slot is reused four changes later; every caller waits for its stream at least once per change.
static int device_list(dpmm_ctx *c, int32_t *dst, std::vector<int32_t> &shadow, const int32_t *data, size_t n, hipStream_t st = nullptr) {
    if (shadow.size() == n && (n == 0 || memcmp(shadow.data(), data, sizeof(int32_t) * n) == 0)) return DPMM_OK;
    shadow.assign(data, data + n);
    if (n == 0) return DPMM_OK;
    const size_t bytes = sizeof(int32_t) * n;
    int32_t *&slot = c->h_list[c->h_list_next];
    size_t &cap = c->h_list_cap[c->h_list_next];
    c->h_list_next = (c->h_list_next + 1) % 4;
    if (bytes > cap) {
        if (slot) HIPCHK(c, hipHostFree(slot));
        slot = nullptr; cap = 0;
        size_t nc = 4096;
        while (nc < bytes) nc *= 2;
        HIPCHK(c, hipHostMalloc((void **)&slot, nc, hipHostMallocDefault));
        cap = nc;
    }
    memcpy(slot, data, bytes);
    HIPCHK(c, launch_copy_bytes(dst, slot, bytes, st ? st : c->stream));
    return DPMM_OK;
}
// device storage for `slots` slots (posterior state, kept across growth) and K clusters (draw scratch)
static int master_capacity(dpmm_ctx *c, int slots, int K) {
    const size_t DP = (size_t)c->ma.DP, stride = (size_t)c->packed_stride;
    if (slots > c->master_slots) {
        int ns = std::max(8, c->master_slots);
        while (ns < slots) ns *= 2;
        ns = std::min(ns, DPMM_MAX_CLUSTERS);
        HIPCHK(c, sync_stream(c, c->stream));
        if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
        c->spec_valid = false; c->noise_valid = false; c->apairs_valid = false;
        double *fac = nullptr, *mean = nullptr, *kap = nullptr, *nu = nullptr, *rows = nullptr;
        HIPCHK(c, hipMalloc(&fac, sizeof(double) * 3 * ns * DP * DP));
        HIPCHK(c, hipMalloc(&mean, sizeof(double) * 3 * ns * DP));
        HIPCHK(c, hipMalloc(&kap, sizeof(double) * 3 * ns));
        HIPCHK(c, hipMalloc(&nu, sizeof(double) * 3 * ns));
        HIPCHK(c, hipMalloc(&rows, sizeof(double) * 2 * ns * stride));
        if (c->master_slots > 0) {
            const size_t os = (size_t)c->master_slots;
            HIPCHK(c, hipMemcpy(fac, c->ma.fac, sizeof(double) * 3 * os * DP * DP, hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(mean, c->ma.mean, sizeof(double) * 3 * os * DP, hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(kap, c->ma.kap, sizeof(double) * 3 * os, hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nu, c->ma.nu, sizeof(double) * 3 * os, hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(rows, c->ma.rows_store, sizeof(double) * 2 * os * stride, hipMemcpyDeviceToDevice));
        }
        hipFree(c->ma.fac); hipFree(c->ma.mean); hipFree(c->ma.kap); hipFree(c->ma.nu); hipFree(c->ma.rows_store);
        c->ma.fac = fac; c->ma.mean = mean; c->ma.kap = kap; c->ma.nu = nu; c->ma.rows_store = rows;
        c->master_slots = ns;
    }
    if (K > c->master_K) {
        int nk = std::max(8, c->master_K);
        while (nk < K) nk *= 2;
        nk = std::min(nk, DPMM_MAX_CLUSTERS);
        HIPCHK(c, sync_stream(c, c->stream));
        if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
        c->spec_valid = false; c->noise_valid = false; c->apairs_valid = false;
        for (int i = 0; i < 2; ++i) {
            hipFree(c->d_Y[i]); hipFree(c->d_ld_sigma[i]); hipFree(c->d_mu_draw[i]);
            c->d_Y[i] = nullptr; c->d_ld_sigma[i] = nullptr; c->d_mu_draw[i] = nullptr;
            HIPCHK(c, hipMalloc(&c->d_Y[i], sizeof(double) * 3 * nk * DP * DP));
            HIPCHK(c, hipMalloc(&c->d_ld_sigma[i], sizeof(float) * 3 * nk));
            HIPCHK(c, hipMalloc(&c->d_mu_draw[i], sizeof(float) * 3 * nk * DP));
        }
        c->draws_on_device = false;
        c->master_K = nk;
    }
    return DPMM_OK;
}

int dpmm_niw_master_setup(dpmm_ctx *c, double kappa, double nu, const double *m, const double *psi) {
    if (!c || !m || !psi) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_NIW) return fail(c, DPMM_EINVAL, "the device master exists for the NIW prior only");
    if (c->D > DPMM_MASTER_MAXD) return fail(c, DPMM_ELIMIT, "D exceeds the device master's limit");
    HIPCHK(c, hipSetDevice(c->device));
    const int D = c->D;
    const size_t T = (size_t)D * (D + 1) / 2;
    std::vector<double> lo(T);
    for (int a = 0; a < D; ++a)
        for (int b = 0; b <= a; ++b) lo[(size_t)a * (a + 1) / 2 + b] = 0.5 * (psi[(size_t)a * D + b] + psi[(size_t)b * D + a]);
    HIPCHK(c, sync_stream(c, c->stream));
    if (!c->d_m0) HIPCHK(c, hipMalloc(&c->d_m0, sizeof(double) * D));
    if (!c->d_psi_lo) HIPCHK(c, hipMalloc(&c->d_psi_lo, sizeof(double) * T));
    HIPCHK(c, hipMemcpy(c->d_m0, m, sizeof(double) * D, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_psi_lo, lo.data(), sizeof(double) * T, hipMemcpyHostToDevice));
    c->ma.D = D; c->ma.DP = 16 * ((D + 15) / 16); c->ma.packed_stride = c->packed_stride;
    c->ma.kappa0 = kappa; c->ma.nu0 = nu; c->ma.m0 = c->d_m0; c->ma.psi_lo = c->d_psi_lo; c->ma.seed = c->seed;
    if (!c->h_draw) HIPCHK(c, hipHostMalloc((void **)&c->h_draw, sizeof(float) * 3 * DPMM_MAX_CLUSTERS, hipHostMallocDefault));
    if (!c->stream2) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_master, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_spec, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_noise, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_pairs, hipEventDisableTiming));
        HIPCHK(c, hipMalloc(&c->d_jobs, sizeof(int32_t) * 2 * DPMM_MAX_CLUSTERS));
        HIPCHK(c, hipMalloc(&c->d_dslots, sizeof(int32_t) * DPMM_MAX_CLUSTERS));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    c->spec_inflight = false; c->spec_valid = false; c->noise_inflight = false; c->noise_valid = false; c->noise_pending = false;
    c->apairs_inflight = false; c->apairs_valid = false;
    c->master = true;
    return DPMM_OK;
}

int dpmm_step_stats_device(dpmm_ctx *c, uint32_t reset_epoch, const uint8_t **bad) {
    if (!c || !bad) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    const size_t out_bytes = sizeof(double) * 2 * (size_t)std::max(c->K, 1) * (size_t)c->packed_stride;
    if (int rc = ensure_out(c, DPMM_MAX_CLUSTERS + 64)) return rc;
    if (int rc = run_stats(c, nullptr, 0, true, reset_epoch)) return rc;
    HIPCHK(c, launch_copy_bytes(c->h_out, reinterpret_cast<const uint8_t *>(c->d_out) + out_bytes, (size_t)c->K + 1, c->stream));
    if (int rc = flush_undo(c)) return rc;
    HIPCHK(c, sync_stream(c, c->stream));
    *bad = reinterpret_cast<const uint8_t *>(c->h_out);
    return DPMM_OK;
}

static const uint64_t MASTER_MARK = 0x7ff8dead0000beefull;
static bool master_records_marked(const double *sm, int64_t nrec) {
    const volatile uint64_t *w = reinterpret_cast<const volatile uint64_t *>(sm);
    for (int64_t r = nrec - 1; r >= 0; --r)           // (from the back: the last workgroups of the launch finish last)
        for (int e = 0; e < 5; ++e) if (w[r * DPMM_MASTER_NSCALARS + e] == MASTER_MARK) return true;
    return false;
}
static bool master_marks_left(const double *sm, int K) { return master_records_marked(sm, 3 * (int64_t)K); }
static void master_mark_records(double *sm, int64_t nrec) {
    for (int64_t r = 0; r < nrec; ++r)
        for (int e = 0; e < 5; ++e) std::memcpy(&sm[r * DPMM_MASTER_NSCALARS + e], &MASTER_MARK, 8);
}
// The wait of dpmm_step_master_device WITHOUT an event (DPMM_OPT_MASTER_POLL, round 6).  The posterior workgroups store their five scalars per
// distribution straight into pinned host memory; every record starts out as MASTER_MARK (a NaN payload no kernel produces), so "no mark left"
// IS "the posteriors' results are on the host" -- the witness of rounds 3-5 promoted to the wait itself.  What it buys: no event record sits
// between the posteriors and the draws launched ahead behind them (a barrier packet with a completion signal: 6 us of stream time at
// N = 1e7, 10 us at the 8-GPU shard size where the second stream's normals join there too), and the host sees the records when they land
// instead of when the runtime has processed the signal.  Everything an EARLIER kernel of the stream wrote to pinned memory (the bad-cluster
// flags) is complete by then: a kernel starts after its predecessor's end-of-kernel release.  The loop keeps the watchdog's contract (a wait
// behind an enqueued collective is armed; an abort ends it) and asks the stream once a millisecond whether it failed or drained.
static int wait_master_records(dpmm_ctx *c, const double *sm, int64_t nrec, const double *pr, int64_t npr) {
    const bool armed = c->watchdog && c->coll_pending;
    if (armed) wd_arm(c);
    int rc = DPMM_OK;
    const auto t0 = std::chrono::steady_clock::now();
    auto next_query = t0 + std::chrono::milliseconds(1);
    std::chrono::steady_clock::time_point drained{};
    bool is_drained = false;
    for (uint64_t spin = 0;; ++spin) {
        if (!master_records_marked(sm, nrec) && !(npr > 0 && master_records_marked(pr, npr))) break;
        __builtin_ia32_pause();
        if ((spin & 63u) != 63u) continue;
        if (c->comm_aborted.load(std::memory_order_relaxed)) break;                         // (reported by wd_disarm below)
        const auto now = std::chrono::steady_clock::now();
        if (now < next_query) continue;
        next_query = now + std::chrono::milliseconds(1);
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) {       // the stream is empty and records are still marked: stores in flight (seen in rounds 3-5 behind an event wait) -- or never written
            if (!is_drained) { is_drained = true; drained = now; ++c->dbg_early_wait; }
            else if (now - drained > std::chrono::seconds(5)) { rc = fail(c, DPMM_EHIP, "posterior records missing 5 s after the stream drained"); break; }
        } else if (q != hipErrorNotReady) { c->err = std::string("hipStreamQuery while waiting for the posteriors: ") + hipGetErrorString(q); rc = DPMM_EHIP; break; }
    }
    if (armed && wd_disarm(c)) {
        c->err = "collective timed out: a peer rank is gone or stuck (communicator aborted after DPMM_OPT_COMM_TIMEOUT_MS)";
        return DPMM_ECOMM;
    }
    if (c->comm_aborted.load(std::memory_order_relaxed) && rc == DPMM_OK && master_records_marked(sm, nrec)) return fail(c, DPMM_ECOMM, "communicator aborted");
    return rc;
}

int dpmm_debug_set_prelaunch_hook(void (*fn)(void *), void *arg) {
    dpmm::g_prelaunch_arg = arg;
    dpmm::g_prelaunch = fn;
    return DPMM_OK;
}

int dpmm_debug_counters(dpmm_ctx *c, int64_t *out, int n) {
    if (!c || !out || n < 1) return DPMM_EINVAL;
    for (int i = 0; i < n; ++i) out[i] = 0;
    out[0] = c->dbg_early_wait;
    return DPMM_OK;
}

int dpmm_step_master_device(dpmm_ctx *c, uint32_t reset_epoch, const int32_t *slots, uint32_t draw_epoch, const uint8_t **bad, const double **small) {
    if (!c || !slots || !bad || !small) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    const int K = c->K;
    if (K < 1) return fail(c, DPMM_ESTATE, "no clusters");
    HIPCHK(c, hipSetDevice(c->device));
    int top = 0;
    for (int k = 0; k < K; ++k) {
        if (slots[k] < 0 || slots[k] >= DPMM_MAX_CLUSTERS) return fail(c, DPMM_EINVAL, "slot out of range");
        top = std::max(top, slots[k] + 1);
    }
    if (int rc = noise_flush(c)) return rc;
    if (int rc = master_capacity(c, top, draw_epoch ? K : 0)) return rc;
    if (int rc = spec_join(c)) return rc;                  // unused early draws of the last call still read the factors
    c->spec_valid = false;
    const size_t out_bytes = sizeof(double) * 2 * (size_t)K * (size_t)c->packed_stride;
    if (int rc = ensure_out(c, DPMM_MAX_CLUSTERS + 64)) return rc;
    const size_t jobs_bytes = (sizeof(int32_t) * 2 * (size_t)K + 63) & ~(size_t)63;
    if (int rc = master_pinned(c, jobs_bytes + sizeof(double) * 3 * DPMM_MASTER_NSCALARS * (size_t)K)) return rc;
    // every user of the pinned block waits for its kernels before returning, so it is free here; no wait -- the sweep is still in flight
    std::vector<int32_t> jobs(2 * (size_t)K);
    for (int k = 0; k < K; ++k) { jobs[2 * k] = k; jobs[2 * k + 1] = slots[k]; }
    if (int rc = device_list(c, c->d_jobs, c->jobs_shadow, jobs.data(), jobs.size())) return rc;
    if (draw_epoch) if (int rc = device_list(c, c->d_dslots, c->dslots_shadow, slots, (size_t)K)) return rc;
    for (int32_t sl : c->apairs_req) if (sl >= top) { c->apairs_req.clear(); break; }      // (a pair of a slot this pass does not cover: no job)
    const int napairs = (int)(c->apairs_req.size() / 2);
    c->apairs_valid = false;
    bool fuse_pairs = true;
    if (napairs > 0) {
        if ((size_t)napairs > c->apairs_cap) {
            HIPCHK(c, sync_stream(c, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream2));
            size_t cap = 64;
            while (cap < (size_t)napairs) cap *= 2;
            hipFree(c->d_apairs); c->d_apairs = nullptr;
            if (c->h_apairs) hipHostFree(c->h_apairs);
            c->h_apairs = nullptr; c->apairs_cap = 0; c->apairs_shadow.clear();
            HIPCHK(c, hipMalloc(&c->d_apairs, sizeof(int32_t) * 2 * cap));
            HIPCHK(c, hipHostMalloc((void **)&c->h_apairs, sizeof(double) * DPMM_MASTER_NSCALARS * cap, hipHostMallocDefault));
            c->apairs_cap = cap;
        }
        if (c->ma.DP > 128 && (size_t)napairs > c->pair_cap) {      // scratch matrices of the large-D pair kernels
            HIPCHK(c, sync_stream(c, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream2));
            hipFree(c->d_pairs); c->d_pairs = nullptr; c->pair_cap = 0;
            size_t cap = 64;
            while (cap < (size_t)napairs) cap *= 2;
            HIPCHK(c, hipMalloc(&c->d_pairs, sizeof(double) * cap * (size_t)c->ma.DP * (size_t)c->ma.DP));
            c->pair_cap = cap;
        }
        // (on the second stream, where the pair job runs: the list changes with the merge gates, and a copy on the main stream would sit
        // between the sweep and the statistics kernels)
        if (niw_master_can_fuse_pairs(c->ma)) {
            // D <= 128: the pair jobs ride in the posteriors' launch (below) and name CLUSTERS of this pass
            std::vector<int32_t> inv((size_t)top, -1), kp(c->apairs_req.size());
            for (int k = 0; k < K; ++k) inv[slots[k]] = k;
            for (size_t i = 0; i < kp.size(); ++i) {
                kp[i] = inv[c->apairs_req[i]];
                if (kp[i] < 0) { fuse_pairs = false; break; }
            }
            if (fuse_pairs) {
                // The fused pair jobs read their two cluster indices straight from pinned host memory (round 6): the list changes whenever a merge
                // gate opens or closes (a bad-cluster reset closes one: most steps of the bench chain), and the copy launch that brought it to the
                // device sat on the main stream between the sweep and the statistics kernels (5 us).  Two buffers take turns: the launch that read
                // the other one has been waited for (its records were) before a new list is written.
                if (c->opt_chain & 4) {
                    if (!(c->apairs_shadow.size() == kp.size() && memcmp(c->apairs_shadow.data(), kp.data(), sizeof(int32_t) * kp.size()) == 0) || !c->apairs_pinned_cur) {
                        c->apairs_shadow.assign(kp.begin(), kp.end());
                        c->apairs_pin_flip ^= 1;
                        int32_t *&buf = c->h_apairs_list[c->apairs_pin_flip];
                        size_t &cap = c->h_apairs_list_cap[c->apairs_pin_flip];
                        if (kp.size() > cap) {
                            if (buf) HIPCHK(c, hipHostFree(buf));
                            buf = nullptr; cap = 0;
                            size_t nc = 2048;
                            while (nc < kp.size()) nc *= 2;
                            HIPCHK(c, hipHostMalloc((void **)&buf, sizeof(int32_t) * nc, hipHostMallocDefault));
                            cap = nc;
                        }
                        memcpy(buf, kp.data(), sizeof(int32_t) * kp.size());
                        c->apairs_pinned_cur = buf;
                    }
                } else {
                    if (c->apairs_pinned_cur) { c->apairs_shadow.clear(); c->apairs_pinned_cur = nullptr; }      // (the shadow described the pinned list, not d_apairs)
                    if (int rc = device_list(c, c->d_apairs, c->apairs_shadow, kp.data(), kp.size())) return rc;
                }
            }
        } else fuse_pairs = false;
        if (!fuse_pairs) {
            if (c->apairs_pinned_cur) { c->apairs_shadow.clear(); c->apairs_pinned_cur = nullptr; }
            if (int rc = device_list(c, c->d_apairs, c->apairs_shadow, c->apairs_req.data(), c->apairs_req.size(), c->stream2)) return rc;
        }
    } else fuse_pairs = false;
    double *sm = reinterpret_cast<double *>(c->h_master + jobs_bytes);
    // Witness of the event wait below: every scalar record the posteriors write starts out as a marker no kernel produces (a NaN with a
    // payload; the kernels' own NaN is the plain quiet one).  A record still carrying it after the wait means the wait returned before
    // the kernels' stores reached host memory: counted (dpmm_debug_counters) and waited out.
    master_mark_records(sm, 3 * (int64_t)K);
    const bool poll = c->opt_master_poll != 0;
    bool flags_sent = false;
    if (int rc = run_stats(c, nullptr, 0, true, reset_epoch, reinterpret_cast<uint8_t *>(c->h_out), &flags_sent)) return rc;
    if (!flags_sent) HIPCHK(c, launch_copy_bytes(c->h_out, reinterpret_cast<const uint8_t *>(c->d_out) + out_bytes, (size_t)K + 1, c->stream));
    // DPMM_OPT_CHAIN_FUSION bit 16 (D <= 128, draws launched ahead): the normals of those draws are generated by extra workgroups of the
    // posteriors' launch (niw_post_both_kernel) -- no kernel beside the sweep on the second stream, no cross-stream wait in front of the draws
    const bool noise_in_post = draw_epoch != 0 && (c->opt_chain & 16) != 0 && niw_master_can_fuse_pairs(c->ma);
    bool normals_by_post = false;
    if (noise_in_post) {
        if (int rc = noise_join(c)) return rc;                 // (a kernel of the other scheme still writing that buffer: rare -- the option was just switched)
        c->noise_valid = false;
    }
    if ((napairs > 0 && fuse_pairs) || noise_in_post) {
        // posteriors + the pooled pair log-determinants the master may ask for (dpmm_niw_master_pairs_ahead) in ONE launch: the pairs need
        // the rows of this pass only.  (On the second stream behind the posteriors they reached the host 12 + 26 us later.)
        const int nfp = fuse_pairs ? napairs : 0;
        if (poll && nfp > 0) master_mark_records(c->h_apairs, nfp);      // (nobody reads the pair records between two passes: dpmm_niw_master_pairs copies them out)
        HIPCHK(c, launch_niw_master_posterior_pairs(c->ma, c->d_jobs, K, c->d_out, sm, ((c->opt_chain & 4) && c->apairs_pinned_cur) ? c->apairs_pinned_cur : c->d_apairs, nfp, c->h_apairs, c->stream,
                                                    noise_in_post ? 3 * K : 0, draw_epoch, noise_in_post ? c->d_Y[c->draw_cur ^ 1] : nullptr));
        normals_by_post = noise_in_post;
    } else HIPCHK(c, launch_niw_master_posterior(c->ma, c->d_jobs, K, c->d_out, sm, c->stream));
    if (napairs > 0 && fuse_pairs) {
        c->apairs_index.clear();
        for (int p = 0; p < napairs; ++p) c->apairs_index[((uint32_t)c->apairs_req[2 * p] << 16) | (uint32_t)c->apairs_req[2 * p + 1]] = p;
        c->apairs_dirty.assign((size_t)c->master_slots, 0);
        c->apairs_inflight = false; c->apairs_valid = true;      // (the host waits for ev_master below: the records are there when it returns)
        c->apairs_req.clear();
    }
    const bool fused_pairs_launched = napairs > 0 && fuse_pairs;
    const bool need_event = !poll || (napairs > 0 && !fuse_pairs);      // (the second stream's pair job waits for the posteriors through it)
    if (need_event) HIPCHK(c, hipEventRecord(c->ev_master, c->stream));
    if (napairs > 0 && !fuse_pairs) {
        // The pooled pair log-determinants the master may ask for after its split decisions (dpmm_niw_master_pairs_ahead): they need the
        // stored rows of this pass only, so they run on the second stream while the host works; dpmm_niw_master_pairs answers from
        // them when every pair it is asked for is among them and none of their slots got new statistics in between.
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_master, 0));
        HIPCHK(c, launch_niw_master_pairs(c->ma, c->d_apairs, napairs, c->d_pairs, c->h_apairs, c->stream2));
        HIPCHK(c, hipEventRecord(c->ev_pairs, c->stream2));
        c->apairs_index.clear();
        for (int p = 0; p < napairs; ++p) c->apairs_index[((uint32_t)c->apairs_req[2 * p] << 16) | (uint32_t)c->apairs_req[2 * p + 1]] = p;
        c->apairs_dirty.assign((size_t)c->master_slots, 0);
        c->apairs_inflight = true; c->apairs_valid = true;
        c->apairs_req.clear();
    }
    if (draw_epoch) {
        // The next parameter draws, launched now on the second stream: they need the posteriors only (not the weights, not the
        // master's split / merge decisions), so they run while the host works on the scalars this call returns.  If nothing changes
        // the cluster -> slot map until dpmm_niw_master_draw(draw_epoch, ...), that call finds them done and launches the hand-over alone.
        // (On the MAIN stream, right behind the posteriors and behind the event the host waits for: a hand-over across two streams costs
        // ~15 us of signalling each way -- posteriors -> draws -> pack took 26 + 14 + 27 + 17 + 10 us at the 8-GPU shard size, the gaps
        // being the two cross-stream waits.  The normals were generated on the second stream during the sweep: that event is long done.)
        const int32_t *hs = c->d_dslots;
        const bool have_normals = normals_by_post || noise_ready(c, draw_epoch, K, c->draw_cur ^ 1);
        if (int rc = noise_join(c)) return rc;             // (also when its normals are not the ones wanted: the kernel writes the buffer the draws go to)
        NiwMasterArgs ma = c->ma;
        ma.mu_draw = c->d_mu_draw[c->draw_cur ^ 1];
        HIPCHK(c, launch_niw_master_draw(ma, hs, K, draw_epoch, c->d_Y[c->draw_cur ^ 1], c->d_ld_sigma[c->draw_cur ^ 1], nullptr, nullptr,
                                         nullptr, nullptr, nullptr, nullptr, c->NB, nullptr, 1 | (have_normals ? 4 : 0), c->stream));
        c->noise_valid = false;
        c->spec_inflight = false; c->spec_valid = true; c->spec_epoch = draw_epoch;
        c->spec_slots.assign(slots, slots + K);
    }
    if (int rc = flush_undo(c)) return rc;      // (behind the posteriors, the event and the draws launched ahead)
    if (poll) { if (int rc = wait_master_records(c, sm, 3 * (int64_t)K, c->h_apairs, fused_pairs_launched ? napairs : 0)) return rc; }
    else HIPCHK(c, sync_event(c, c->ev_master));
    if (master_marks_left(sm, K)) {
        ++c->dbg_early_wait;
        const auto t0 = std::chrono::steady_clock::now();
        while (master_marks_left(sm, K)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) return fail(c, DPMM_EHIP, "posterior records missing 5 s after the event wait returned");
            HIPCHK(c, sync_stream(c, c->stream));
        }
    }
    c->handover_inflight = false;          // (recorded behind the last hand-over kernel on the same stream)
    *bad = reinterpret_cast<const uint8_t *>(c->h_out);
    *small = sm;
    return DPMM_OK;
}

int dpmm_suffstats_device(dpmm_ctx *c, const int64_t *idx, int n_idx) {
    if (!c) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    return run_stats(c, idx, n_idx);
}

int dpmm_niw_master_posterior(dpmm_ctx *c, const int64_t *clusters, const int32_t *slots, int n, const double **small) {
    if (!c || !slots || !small || n < 0) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    if (n == 0) { *small = nullptr; return DPMM_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    int top = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t k = clusters ? clusters[i] : i + 1;
        if (k < 1 || k > c->K) return fail(c, DPMM_EINVAL, "cluster index out of range");
        if (slots[i] < 0 || slots[i] >= DPMM_MAX_CLUSTERS) return fail(c, DPMM_EINVAL, "slot out of range");
        top = std::max(top, slots[i] + 1);
    }
    if (int rc = master_capacity(c, top, 0)) return rc;
    if (int rc = spec_join(c)) return rc;
    c->spec_valid = false;                                 // some factors change: draws launched ahead are not the ones to use
    for (int i = 0; i < n; ++i) if ((size_t)slots[i] < c->apairs_dirty.size()) c->apairs_dirty[slots[i]] = 1;      // ... nor pooled pairs that involve these slots
    const size_t jobs_bytes = (sizeof(int32_t) * 2 * (size_t)n + 63) & ~(size_t)63;
    if (int rc = master_pinned(c, jobs_bytes + sizeof(double) * 3 * DPMM_MASTER_NSCALARS * (size_t)n)) return rc;
    HIPCHK(c, sync_stream(c, c->stream));            // nobody reads the pinned block any more
    std::vector<int32_t> jobs(2 * (size_t)n);
    for (int i = 0; i < n; ++i) { jobs[2 * i] = (int32_t)((clusters ? clusters[i] : i + 1) - 1); jobs[2 * i + 1] = slots[i]; }
    if (int rc = device_list(c, c->d_jobs, c->jobs_shadow, jobs.data(), jobs.size())) return rc;
    double *sm = reinterpret_cast<double *>(c->h_master + jobs_bytes);
    HIPCHK(c, launch_niw_master_posterior(c->ma, c->d_jobs, n, c->d_out, sm, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    *small = sm;
    return DPMM_OK;
}

int dpmm_niw_master_draw(dpmm_ctx *c, uint32_t epoch, int K, const int32_t *slot_of_cluster, const float *lr, const float *w) {
    if (!c || !slot_of_cluster || !lr || !w) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    if (int rc = check_K(c, K)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    for (int k = 0; k < K; ++k)
        if (slot_of_cluster[k] < 0 || slot_of_cluster[k] >= c->master_slots) return fail(c, DPMM_EINVAL, "slot without a posterior on the device");
    if (int rc = noise_flush(c)) return rc;
    if (int rc = ensure_capacity(c, K)) return rc;
    if (int rc = master_capacity(c, 0, K)) return rc;
    // small inputs through a pinned block of their own (the posterior's block may still be read by its caller)
    // lr / w through a pinned block of their own (h_draw; the shared staging h_pin belongs to calls that synchronise before they use it).
    // The previous hand-over kernel reads lr / w from this block: wait only if the host has not waited for anything behind it since
    // (in the engine's loop it has -- for the posteriors of the step -- and the stream now carries the draws launched ahead, which a
    // synchronise here would wait out: 27 us on the host's critical path at D = 64)
    if (c->handover_inflight) HIPCHK(c, sync_stream(c, c->stream));
    c->handover_inflight = true;
    float *hlr = reinterpret_cast<float *>(c->h_draw), *hw = hlr + 2 * K;
    memcpy(hlr, lr, sizeof(float) * 2 * K);
    memcpy(hw, w, sizeof(float) * K);
    c->have_tail = c->opt_tail && c->D >= 4 && c->D % 4 == 0 && K > 1;
    // draws launched ahead by dpmm_step_master_device for this epoch and this cluster -> slot map: only the hand-over is left
    const bool ahead = c->spec_valid && c->spec_epoch == epoch && (int)c->spec_slots.size() == K &&
                       memcmp(c->spec_slots.data(), slot_of_cluster, sizeof(int32_t) * K) == 0;
    if (int rc = spec_join(c)) return rc;
    c->spec_valid = false;
    if (int rc = noise_join(c)) return rc;                 // (no-op when the draws launched ahead already waited for the normals)
    if (!ahead) if (int rc = device_list(c, c->d_dslots, c->dslots_shadow, slot_of_cluster, (size_t)K)) return rc;
    const int32_t *hs = c->d_dslots;
    c->draw_cur ^= 1;
    NiwMasterArgs ma = c->ma;
    ma.mu_draw = c->d_mu_draw[c->draw_cur];
    const bool normals = !ahead && noise_ready(c, epoch, K, c->draw_cur);       // (noise_join above made the main stream wait for them)
    c->predictive = false;              // (before the tables, as in set_params)
    // DPMM_OPT_CHAIN_FUSION bit 2: the hand-over partitioned by role, with the three-plane images in the same launch (no niw_b3_pack launch)
    const bool roles = (c->opt_chain & 2) != 0, b3_in_pack = roles && want_b3_images(c), pb_in_pack = b3_in_pack && want_pair_ball(c, K);
    HIPCHK(c, launch_niw_master_draw(ma, hs, K, epoch, c->d_Y[c->draw_cur], c->d_ld_sigma[c->draw_cur], hlr, hw, c->d_Rp, c->d_mup, c->d_cst,
                                     c->have_tail ? c->d_tail : nullptr, c->NB, c->d_work, (ahead ? 2 : (normals ? 7 : 3)) | (roles ? 8 : 0) | (b3_in_pack ? 16 : 0) | (pb_in_pack ? 32 : 0), c->stream));
    if (int rc = direction_tables(c, K, b3_in_pack, pb_in_pack)) return rc;
    // The normals of the NEXT draws (epoch + 1, a few clusters more than now for the splits in between) into the other buffer, on the
    // second stream: they depend on nothing the master decides, and run beside the sweep instead of in front of it.  Whoever draws
    // with another epoch, or for more clusters, generates its own.
    // (launched by noise_flush right AFTER the sweep kernel of this step: the launch and its event record would otherwise sit on the
    // host's critical path between the master's decisions and the sweep launch)
    c->noise_pending = c->opt_noise_ahead < 0 ? (c->D >= 128 || c->n < 4000000) : c->opt_noise_ahead != 0; c->noise_pend_epoch = epoch + 1; c->noise_pend_nmat = 3 * std::min(K + 4, c->master_K);
    if ((c->opt_chain & 16) && niw_master_can_fuse_pairs(c->ma)) c->noise_pending = false;      // (the posteriors' launch generates them: dpmm_step_master_device)
    c->work_zeroed = true;
    c->have_screen_prep = false;     // (the K > 64 far mask needs the raw factors: not built on this path; the tail screen is)
    c->K = K;
    c->have_params = true;
    c->predictive = false;
    c->draws_on_device = true;
    return DPMM_OK;
}

int dpmm_niw_master_pairs_ahead(dpmm_ctx *c, const int32_t *slots_i, const int32_t *slots_j, int n) {
    if (!c || n < 0 || (n > 0 && (!slots_i || !slots_j))) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    c->apairs_req.clear();
    for (int i = 0; i < n; ++i) {
        if (slots_i[i] < 0 || slots_i[i] >= DPMM_MAX_CLUSTERS || slots_j[i] < 0 || slots_j[i] >= DPMM_MAX_CLUSTERS) { c->apairs_req.clear(); return fail(c, DPMM_EINVAL, "slot out of range"); }
        c->apairs_req.push_back(slots_i[i]); c->apairs_req.push_back(slots_j[i]);
    }
    return DPMM_OK;
}

int dpmm_niw_master_pairs(dpmm_ctx *c, const int32_t *slots_i, const int32_t *slots_j, int n, const double **small) {
    if (!c || !slots_i || !slots_j || !small || n < 0) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    if (n == 0) { *small = nullptr; return DPMM_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < n; ++i)
        if (slots_i[i] < 0 || slots_i[i] >= c->master_slots || slots_j[i] < 0 || slots_j[i] >= c->master_slots) return fail(c, DPMM_EINVAL, "slot out of range");
    const size_t DP = (size_t)c->ma.DP;
    if (c->apairs_valid) {      // launched ahead with the posteriors?
        bool all = true;
        std::vector<int> rec((size_t)n);
        for (int i = 0; i < n && all; ++i) {
            const auto it = c->apairs_index.find(((uint32_t)slots_i[i] << 16) | (uint32_t)slots_j[i]);
            all = it != c->apairs_index.end() && !c->apairs_dirty[slots_i[i]] && !c->apairs_dirty[slots_j[i]];
            if (all) rec[i] = it->second;
        }
        if (all) {
            if (int rc = master_pinned(c, sizeof(double) * DPMM_MASTER_NSCALARS * (size_t)n)) return rc;
            HIPCHK(c, hipEventSynchronize(c->ev_pairs));
            double *sm = reinterpret_cast<double *>(c->h_master);      // (free: every user waits for its kernels before it returns)
            for (int i = 0; i < n; ++i) memcpy(sm + (size_t)i * DPMM_MASTER_NSCALARS, c->h_apairs + (size_t)rec[i] * DPMM_MASTER_NSCALARS, sizeof(double) * DPMM_MASTER_NSCALARS);
            *small = sm;
            return DPMM_OK;
        }
    }
    if (int rc = spec_join(c)) return rc;          // (a job launched ahead may still use the scratch matrices)
    if ((size_t)n > c->pair_cap) {
        HIPCHK(c, sync_stream(c, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream2));
        hipFree(c->d_pairs); c->d_pairs = nullptr; c->pair_cap = 0;
        size_t cap = 64;
        while (cap < (size_t)n) cap *= 2;
        HIPCHK(c, hipMalloc(&c->d_pairs, sizeof(double) * cap * DP * DP));
        c->pair_cap = cap;
    }
    const size_t idx_bytes = (sizeof(int32_t) * 2 * (size_t)n + 63) & ~(size_t)63;
    if (int rc = master_pinned(c, idx_bytes + sizeof(double) * DPMM_MASTER_NSCALARS * (size_t)n)) return rc;
    HIPCHK(c, sync_stream(c, c->stream));
    int32_t *pr = reinterpret_cast<int32_t *>(c->h_master);
    for (int i = 0; i < n; ++i) { pr[2 * i] = slots_i[i]; pr[2 * i + 1] = slots_j[i]; }
    double *sm = reinterpret_cast<double *>(c->h_master + idx_bytes);
    HIPCHK(c, launch_niw_master_pairs(c->ma, pr, n, c->d_pairs, sm, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    *small = sm;
    return DPMM_OK;
}

int dpmm_niw_master_put_rows(dpmm_ctx *c, const double *rows, int K) {
    if (!c || !rows) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    if (int rc = check_K(c, K)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_capacity(c, K)) return rc;
    HIPCHK(c, sync_stream(c, c->stream));
    HIPCHK(c, hipMemcpy(c->d_out, rows, sizeof(double) * 2 * (size_t)K * (size_t)c->packed_stride, hipMemcpyHostToDevice));
    c->apairs_valid = false;
    if (K != c->K) c->have_params = false;
    c->K = K;
    return DPMM_OK;
}

int dpmm_niw_master_rows(dpmm_ctx *c, const int32_t *slots, int n, double *out) {
    if (!c || !slots || !out || n < 0) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    if (n == 0) return DPMM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t stride = (size_t)c->packed_stride, bytes = sizeof(double) * 2 * (size_t)n * stride;
    for (int i = 0; i < n; ++i)
        if (slots[i] < 0 || slots[i] >= c->master_slots) return fail(c, DPMM_EINVAL, "slot out of range");
    // one gather kernel into pinned memory (n separate copies cost 6 us each), then a host copy to the caller's buffer
    if (int rc = ensure_out(c, bytes)) return rc;
    if (int rc = ensure_pinned(c, sizeof(int32_t) * (size_t)n + 64)) return rc;
    HIPCHK(c, sync_stream(c, c->stream));
    memcpy(c->h_pin, slots, sizeof(int32_t) * (size_t)n);
    HIPCHK(c, launch_niw_rows_gather(c->ma.rows_store, reinterpret_cast<const int32_t *>(c->h_pin), n, (int64_t)stride,
                                     reinterpret_cast<double *>(c->h_out), c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    memcpy(out, c->h_out, bytes);
    return DPMM_OK;
}

int dpmm_niw_master_draws(dpmm_ctx *c, int K, float *mu, float *R, float *logdet) {
    if (!c || !mu || !R || !logdet) return DPMM_EINVAL;
    if (!c->master || !c->draws_on_device || K != c->K) return fail(c, DPMM_ESTATE, "no device draws for this K");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_stream(c, c->stream));
    const size_t D = (size_t)c->D, DP = (size_t)c->ma.DP;
    std::vector<double> Y(3 * (size_t)K * DP * DP);
    std::vector<float> m(3 * (size_t)K * DP);
    HIPCHK(c, hipMemcpy(Y.data(), c->d_Y[c->draw_cur], sizeof(double) * Y.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(m.data(), c->d_mu_draw[c->draw_cur], sizeof(float) * m.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(logdet, c->d_ld_sigma[c->draw_cur], sizeof(float) * 3 * (size_t)K, hipMemcpyDeviceToHost));
    for (size_t j = 0; j < 3 * (size_t)K; ++j) {
        for (size_t d = 0; d < D; ++d) mu[j * D + d] = m[j * DP + d];
        for (size_t r = 0; r < D; ++r)
            for (size_t cc = 0; cc < D; ++cc) R[(j * D + r) * D + cc] = cc >= r ? (float)Y[(j * DP + cc) * DP + r] : 0.f;
    }
    return DPMM_OK;
}

// ---- the Multinomial master's draws on the device ----------------------------------------------------------------------------
int dpmm_mult_master_setup(dpmm_ctx *c, const float *alpha, const float *alpha_outlier) {
    if (!c || !alpha) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_MULT) return fail(c, DPMM_EINVAL, "dpmm_mult_master_setup: the context is not Multinomial");
    if (c->D > DPMM_MULT_MASTER_MAXD) return fail(c, DPMM_ELIMIT, "D exceeds the device Dirichlet draw's limit");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_stream(c, c->stream));
    if (!c->d_malpha) HIPCHK(c, hipMalloc(&c->d_malpha, sizeof(float) * 2 * (size_t)c->ldx));
    HIPCHK(c, hipMemsetAsync(c->d_malpha, 0, sizeof(float) * 2 * (size_t)c->ldx, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    HIPCHK(c, hipMemcpy(c->d_malpha, alpha, sizeof(float) * (size_t)c->D, hipMemcpyHostToDevice));
    if (alpha_outlier) HIPCHK(c, hipMemcpy(c->d_malpha + c->ldx, alpha_outlier, sizeof(float) * (size_t)c->D, hipMemcpyHostToDevice));
    c->mult_has_alpha1 = alpha_outlier != nullptr;
    for (int p = 0; p < 2; ++p) {
        const float *al = p ? alpha_outlier : alpha;
        double sa = 0.0, sl = 0.0;
        if (al) for (int d = 0; d < c->D; ++d) { sa += (double)al[d]; sl += lgamma((double)al[d]); }
        c->mult_prior_c[2 * p] = sa; c->mult_prior_c[2 * p + 1] = sl;
    }
    if (!c->d_mpairs) HIPCHK(c, hipMalloc(&c->d_mpairs, sizeof(int32_t) * 2 * DPMM_MULT_MASTER_MAXPAIRS));
    if (!c->h_marg) HIPCHK(c, hipHostMalloc((void **)&c->h_marg, sizeof(double) * (6 * DPMM_MAX_CLUSTERS + DPMM_MULT_MASTER_MAXPAIRS), hipHostMallocDefault));
    c->marg_valid = false; c->marg_req = false; c->mpairs_shadow.clear();
    if (!c->h_draw) HIPCHK(c, hipHostMalloc((void **)&c->h_draw, sizeof(float) * 3 * DPMM_MAX_CLUSTERS, hipHostMallocDefault));
    c->mult_master = true;
    return DPMM_OK;
}

int dpmm_mult_master_put_rows(dpmm_ctx *c, const double *rows, int K) {
    if (!c || !rows) return DPMM_EINVAL;
    if (!c->mult_master) return fail(c, DPMM_ESTATE, "dpmm_mult_master_setup first");
    if (int rc = check_K(c, K)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_capacity(c, K)) return rc;
    HIPCHK(c, sync_stream(c, c->stream));
    HIPCHK(c, hipMemcpy(c->d_out, rows, sizeof(double) * 2 * (size_t)K * (size_t)c->packed_stride, hipMemcpyHostToDevice));
    c->K = K;
    c->rows_full_K = K;
    c->marg_valid = false;
    c->mspec_valid = false;
    c->cache_force = true;            // (the rows did not come from this context's labels: nothing to derive from)
    return DPMM_OK;
}

int dpmm_mult_master_draw(dpmm_ctx *c, uint32_t epoch, int K, int outlier_first, const float *lr, const float *w) {
    if (!c || !lr || !w) return DPMM_EINVAL;
    if (!c->mult_master) return fail(c, DPMM_ESTATE, "dpmm_mult_master_setup first");
    if (int rc = check_K(c, K)) return rc;
    if (c->rows_full_K != K) return fail(c, DPMM_ESTATE, "dpmm_mult_master_draw: the last statistics pass did not leave the rows of all K clusters");
    if (outlier_first && !c->mult_has_alpha1) return fail(c, DPMM_ESTATE, "dpmm_mult_master_draw: no outlier prior was set up");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = ensure_capacity(c, K)) return rc;
    // cst[3k] = log w_k, cst[3k+1+s] = log lr[k][s] through the pinned block.  The block is re-used once the copy of the PREVIOUS call has read
    // it -- known from the host's waits since, not by waiting for the stream here: the stream may be carrying the draws launched ahead and the late
    // rows, and waiting for those would put them back on the host's path
    if (c->cst_inflight && c->cst_gen == c->sync_gen) HIPCHK(c, sync_stream(c, c->stream));     // (never in the engine's loop: dpmm_step_stats waited in between)
    float *hcst = reinterpret_cast<float *>(c->h_draw);
    for (int k = 0; k < K; ++k) { hcst[3 * k] = logf(w[k]); hcst[3 * k + 1] = logf(lr[2 * k]); hcst[3 * k + 2] = logf(lr[2 * k + 1]); }
    HIPCHK(c, launch_copy_bytes(c->d_cst, hcst, sizeof(float) * 3 * K, c->stream));
    c->cst_inflight = true; c->cst_gen = c->sync_gen;
    if (c->mspec_valid && c->mspec_epoch == epoch && c->mspec_K == K && c->mspec_outlier == (outlier_first ? 1 : 0)) {
        // made behind the statistics pass (dpmm_step_stats) from these very rows: the two sets of buffers change places
        std::swap(c->d_raw, c->d_raw2); std::swap(c->d_Rp, c->d_Rp2); std::swap(c->d_Lp16, c->d_Lp16_2); std::swap(c->rp_current, c->rp2_current);
        c->mspec_used += 1;
    } else {
        HIPCHK(c, launch_mult_dirichlet(c->d_out, c->packed_stride, c->d_malpha, c->mult_has_alpha1 ? c->d_malpha + c->ldx : nullptr, outlier_first, c->D, c->ldx,
                                        K, c->seed, epoch, c->d_raw, c->stream));
        c->rp_current = !(c->x_u8 || c->x_bf16_exact);
        if (c->rp_current) HIPCHK(c, launch_mult_pack(c->d_raw, c->d_Rp, 3 * K, c->ldx, c->stream));
        if (c->x_u8) HIPCHK(c, launch_mult_pack_u8(c->d_raw, c->d_Lp16, 3 * K, c->ldx, c->ld8, c->stream));
        else if (c->x_bf16_exact) HIPCHK(c, launch_mult_pack_bf16(c->d_raw, c->d_Lp16, 3 * K, c->ldx, c->stream));
    }
    c->mspec_valid = false;
    c->mdraw_epoch = epoch; c->mdraw_seen = true;
    c->K = K;
    c->have_params = true;
    c->predictive = false;
    c->draws_on_device = true;
    return DPMM_OK;
}

int dpmm_mult_master_rows_on_demand(dpmm_ctx *c, int on) {
    if (!c) return DPMM_EINVAL;
    if (!c->mult_master) return fail(c, DPMM_ESTATE, "dpmm_mult_master_setup first");
    c->rows_on_demand = on != 0;
    return DPMM_OK;
}

int dpmm_mult_master_rows_wait(dpmm_ctx *c) {
    if (!c) return DPMM_EINVAL;
    if (c->rows_late) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, sync_stream(c, c->stream));
        c->rows_late = false;
    }
    return DPMM_OK;
}

int dpmm_debug_mult_draws_ahead(dpmm_ctx *c, long long *used) {
    if (!c || !used) return DPMM_EINVAL;
    *used = c->mspec_used;
    return DPMM_OK;
}

int dpmm_mult_master_draws(dpmm_ctx *c, int K, float *logp) {
    if (!c || !logp) return DPMM_EINVAL;
    if (!c->mult_master || !c->draws_on_device || K != c->K) return fail(c, DPMM_ESTATE, "no device draws for this K");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_stream(c, c->stream));
    HIPCHK(c, hipMemcpy2D(logp, sizeof(float) * (size_t)c->D, c->d_raw, sizeof(float) * (size_t)c->ldx, sizeof(float) * (size_t)c->D, 3 * (size_t)K,
                          hipMemcpyDeviceToHost));
    return DPMM_OK;
}

// the log-marginal kernel over the rows of the last full pass + the pairs asked for (asynchronous; results in the pinned block)
static int mult_marginals_launch(dpmm_ctx *c) {
    c->marg_req = false;
    c->marg_behind_ev = false;
    const int K = c->K;
    if (!c->mult_master || c->rows_full_K != K) return DPMM_OK;           // (nothing to answer from: dpmm_mult_master_marginals will say so)
    if (c->marg_req_outlier && !c->mult_has_alpha1) return DPMM_OK;
    int np = (int)(c->mpairs_req.size() / 2);
    for (int32_t k : c->mpairs_req) if (k < 0 || k >= K) { np = 0; break; }
    if (np > 0) if (int rc = device_list(c, c->d_mpairs, c->mpairs_shadow, c->mpairs_req.data(), 2 * (size_t)np)) return rc;
    HIPCHK(c, launch_mult_marginals(c->d_out, c->packed_stride, c->d_malpha, c->mult_has_alpha1 ? c->d_malpha + c->ldx : nullptr, c->marg_req_outlier,
                                    c->D, K, c->d_mpairs, np, c->mult_prior_c, c->h_marg, c->stream));
    c->marg_valid = true; c->marg_K = K; c->marg_np = np;
    return DPMM_OK;
}

int dpmm_mult_master_pairs_ahead(dpmm_ctx *c, int outlier_first, const int32_t *ki, const int32_t *kj, int n) {
    if (!c || n < 0 || (n > 0 && (!ki || !kj))) return DPMM_EINVAL;
    if (!c->mult_master) return fail(c, DPMM_ESTATE, "dpmm_mult_master_setup first");
    c->mpairs_req.clear();
    if (n <= DPMM_MULT_MASTER_MAXPAIRS)
        for (int p = 0; p < n; ++p) { c->mpairs_req.push_back(ki[p]); c->mpairs_req.push_back(kj[p]); }
    c->marg_req = true; c->marg_req_outlier = outlier_first ? 1 : 0;
    return DPMM_OK;
}

int dpmm_mult_master_marginals(dpmm_ctx *c, int K, const double **rows_nl, const double **pairs_l, int *npairs) {
    if (!c || !rows_nl || !pairs_l || !npairs) return DPMM_EINVAL;
    if (!c->mult_master) return fail(c, DPMM_ESTATE, "dpmm_mult_master_setup first");
    if (int rc = check_K(c, K)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (!(c->marg_valid && c->marg_K == K && c->rows_full_K == K)) {
        if (c->rows_full_K != K || K != c->K) return fail(c, DPMM_ESTATE, "dpmm_mult_master_marginals: the last statistics pass did not leave the rows of all K clusters");
        c->marg_valid = false;
        if (int rc = mult_marginals_launch(c)) return rc;
        if (!c->marg_valid) return fail(c, DPMM_ESTATE, "dpmm_mult_master_marginals: no outlier prior was set up");
    }
    if (c->marg_behind_ev) HIPCHK(c, sync_event(c, c->ev_rows));      // (dpmm_step_stats waited for it already; the stream still carries the draws launched ahead)
    else HIPCHK(c, sync_stream(c, c->stream));        // (no-op behind dpmm_step_stats, which waited for the kernel with the rows)
    *rows_nl = c->h_marg; *pairs_l = c->h_marg + 6 * (size_t)K; *npairs = c->marg_np;
    return DPMM_OK;
}

int dpmm_debug_niw_draw_inputs(dpmm_ctx *c, uint32_t epoch, int K, const int32_t *slot_of_cluster, double *A, double *xi) {
    if (!c || !slot_of_cluster || !A || !xi) return DPMM_EINVAL;
    if (!c->master) return fail(c, DPMM_ESTATE, "dpmm_niw_master_setup first");
    if (K < 1 || K > DPMM_MAX_CLUSTERS) return fail(c, DPMM_EINVAL, "bad K");
    HIPCHK(c, hipSetDevice(c->device));
    for (int k = 0; k < K; ++k)
        if (slot_of_cluster[k] < 0 || slot_of_cluster[k] >= c->master_slots) return fail(c, DPMM_EINVAL, "slot without a posterior on the device");
    const size_t D = (size_t)c->D, nA = 3 * (size_t)K * D * D, nx = 3 * (size_t)K * D;
    double *dA = nullptr, *dxi = nullptr;
    int32_t *dsl = nullptr;
    HIPCHK(c, sync_stream(c, c->stream));
    hipError_t e = hipMalloc(&dA, sizeof(double) * nA);
    if (e == hipSuccess) e = hipMalloc(&dxi, sizeof(double) * nx);
    if (e == hipSuccess) e = hipMalloc(&dsl, sizeof(int32_t) * K);
    if (e == hipSuccess) e = hipMemsetAsync(dA, 0, sizeof(double) * nA, c->stream);
    if (e == hipSuccess) e = hipMemcpy(dsl, slot_of_cluster, sizeof(int32_t) * K, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_niw_draw_inputs(c->ma, dsl, K, epoch, dA, dxi, c->stream);
    if (e == hipSuccess) e = sync_stream(c, c->stream);
    if (e == hipSuccess) e = hipMemcpy(A, dA, sizeof(double) * nA, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(xi, dxi, sizeof(double) * nx, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dxi); hipFree(dsl);
    if (e != hipSuccess) { c->err = std::string("dpmm_debug_niw_draw_inputs: ") + hipGetErrorString(e); return DPMM_EHIP; }
    return DPMM_OK;
}

int dpmm_suffstats_packed(dpmm_ctx *c, const int64_t *idx, int n_idx, double *out) {
    if (!c || !out) return DPMM_EINVAL;
    const double *pk = nullptr;
    if (int rc = dpmm_suffstats_host(c, idx, n_idx, &pk)) return rc;
    memcpy(out, pk, sizeof(double) * 2 * c->K * (size_t)c->packed_stride);
    return DPMM_OK;
}

int dpmm_unpack_suffstats(const dpmm_ctx *c, int K, const double *packed, double *N, double *sum, double *S) {
    if (!c || !packed || !N || !sum) return DPMM_EINVAL;
    const int D = c->D;
    const int64_t ps = c->packed_stride;
    for (int k = 0; k < K; ++k) {
        const double *l = packed + (size_t)(2 * k) * ps, *r = packed + (size_t)(2 * k + 1) * ps;
        N[3 * k + 1] = l[0]; N[3 * k + 2] = r[0]; N[3 * k] = l[0] + r[0];
        for (int d = 0; d < D; ++d) {
            sum[(size_t)(3 * k + 1) * D + d] = l[1 + d];
            sum[(size_t)(3 * k + 2) * D + d] = r[1 + d];
            sum[(size_t)(3 * k) * D + d] = l[1 + d] + r[1 + d];
        }
        if (S && c->prior == DPMM_PRIOR_NIW) {
            double *Sc = S + (size_t)(3 * k) * D * D, *Sl = Sc + (size_t)D * D, *Sr = Sl + (size_t)D * D;
            for (int a = 0; a < D; ++a)
                for (int b = 0; b <= a; ++b) {
                    const double vl = l[1 + D + (size_t)a * (a + 1) / 2 + b], vr = r[1 + D + (size_t)a * (a + 1) / 2 + b];
                    Sl[(size_t)a * D + b] = Sl[(size_t)b * D + a] = vl;
                    Sr[(size_t)a * D + b] = Sr[(size_t)b * D + a] = vr;
                    Sc[(size_t)a * D + b] = Sc[(size_t)b * D + a] = vl + vr;
                }
        }
    }
    return DPMM_OK;
}

static int upload_idx(dpmm_ctx *c, const int64_t *a, const int64_t *b, int n, int limit) {
    if (n < 0 || n > 2 * DPMM_MAX_CLUSTERS) return fail(c, DPMM_ELIMIT, "index list too long");
    std::vector<int32_t> h((size_t)(b ? 2 : 1) * n);
    for (int j = 0; j < n; ++j) {
        if (a[j] < 1 || a[j] > limit) return fail(c, DPMM_EINVAL, "cluster index out of range");
        h[j] = (int32_t)(a[j] - 1);
        if (b) {
            if (b[j] < 1 || b[j] > limit) return fail(c, DPMM_EINVAL, "cluster index out of range");
            h[n + j] = (int32_t)(b[j] - 1);
        }
    }
    if (int rc = ensure_pinned(c, sizeof(int32_t) * h.size())) return rc;
    HIPCHK(c, sync_stream(c, c->stream));
    memcpy(c->h_pin, h.data(), sizeof(int32_t) * h.size());
    HIPCHK(c, launch_copy_bytes(c->d_small, c->h_pin, sizeof(int32_t) * h.size(), c->stream));
    return DPMM_OK;
}

int dpmm_split(dpmm_ctx *c, const int64_t *idx, const int64_t *new_idx, int n, uint32_t epoch) {
    if (!c) return DPMM_EINVAL;
    if (!c->have_labels) return fail(c, DPMM_ESTATE, "labels not initialised");
    if (n == 0) return DPMM_OK;
    if (!idx || !new_idx) return fail(c, DPMM_EINVAL, "null index list");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_idx(c, idx, new_idx, n, DPMM_MAX_CLUSTERS)) return rc;
    if (c->n > 0) HIPCHK(c, launch_split(c->dbins, c->n, c->first, c->d_small, n, c->seed, epoch, c->stream));
    c->rows_full_K = -1;              // (the rows of the last pass no longer describe these labels)
    return DPMM_OK;
}

int dpmm_merge(dpmm_ctx *c, const int64_t *idx, const int64_t *new_idx, int n) {
    if (!c) return DPMM_EINVAL;
    if (!c->have_labels) return fail(c, DPMM_ESTATE, "labels not initialised");
    if (n == 0) return DPMM_OK;
    if (!idx || !new_idx) return fail(c, DPMM_EINVAL, "null index list");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_idx(c, idx, new_idx, n, DPMM_MAX_CLUSTERS)) return rc;
    if (c->n > 0) HIPCHK(c, launch_merge(c->dbins, c->n, c->d_small, n, c->stream));
    c->rows_full_K = -1;
    return DPMM_OK;
}

int dpmm_remove_empty(dpmm_ctx *c, const int64_t *pts_count, int K) {
    if (!c) return DPMM_EINVAL;
    if (!c->have_labels) return fail(c, DPMM_ESTATE, "labels not initialised");
    if (!pts_count || K < 1 || K > DPMM_MAX_CLUSTERS) return fail(c, DPMM_EINVAL, "bad pts_count / K");
    HIPCHK(c, hipSetDevice(c->device));
    // literal simulation of remove_empty_clusters_worker! on the identity label vector -> lookup table
    std::vector<int32_t> map(DPMM_MAX_CLUSTERS);
    for (int l = 1; l <= DPMM_MAX_CLUSTERS; ++l) {
        int v = l, removed = 0;
        for (int k = 1; k <= K; ++k)
            if (pts_count[k - 1] == 0) {
                if (v > k - removed) v -= 1;
                removed += 1;
            }
        map[l - 1] = v - 1;
    }
    if (int rc = ensure_pinned(c, sizeof(int32_t) * map.size())) return rc;
    HIPCHK(c, sync_stream(c, c->stream));
    memcpy(c->h_pin, map.data(), sizeof(int32_t) * map.size());
    HIPCHK(c, launch_copy_bytes(c->d_small, c->h_pin, sizeof(int32_t) * map.size(), c->stream));
    if (c->n > 0) HIPCHK(c, launch_remap(c->dbins, c->n, c->d_small, c->stream));
    c->rows_full_K = -1;
    return DPMM_OK;
}

int dpmm_reset_sublabels(dpmm_ctx *c, const int64_t *idx, int n, uint32_t epoch) {
    if (!c) return DPMM_EINVAL;
    if (!c->have_labels) return fail(c, DPMM_ESTATE, "labels not initialised");
    HIPCHK(c, hipSetDevice(c->device));
    if (idx) {
        if (n == 0) return DPMM_OK;
        if (int rc = upload_idx(c, idx, nullptr, n, DPMM_MAX_CLUSTERS)) return rc;
    }
    if (c->n > 0) HIPCHK(c, launch_reset_sub(c->dbins, c->n, c->first, idx ? c->d_small : nullptr, idx ? n : 0, c->seed, epoch, c->stream));
    c->rows_full_K = -1;
    return DPMM_OK;
}

static int smart_buffers(dpmm_ctx *c) {
    if (c->d_proj) return DPMM_OK;
    const size_t n = (size_t)std::max<int64_t>(c->n, 1);
    HIPCHK(c, hipMalloc(&c->d_proj, sizeof(double) * n));
    HIPCHK(c, hipMalloc(&c->d_vals, sizeof(double) * n));
    HIPCHK(c, hipMalloc(&c->d_smart, sizeof(double) * (4 * (size_t)smart_groups() + 8 + 2 * (size_t)c->D)));
    return DPMM_OK;
}
static int smart_check(dpmm_ctx *c, int64_t cluster) {
    if (!c->have_points || !c->have_labels) return fail(c, DPMM_ESTATE, "smart split needs points and labels");
    if (cluster < 1 || cluster > DPMM_MAX_CLUSTERS) return fail(c, DPMM_EINVAL, "cluster index out of range");
    return DPMM_OK;
}

int dpmm_smart_project(dpmm_ctx *c, int64_t cluster, const double *v, const double *mu, double *values, int64_t *count) {
    if (!c || !v || !mu || !count) return DPMM_EINVAL;
    if (int rc = smart_check(c, cluster)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = smart_buffers(c)) return rc;
    *count = 0;
    if (c->n == 0) return DPMM_OK;
    double *d_v = c->d_smart + 4 * (size_t)smart_groups() + 8, *d_mu = d_v + c->D;
    unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(c->d_smart + 4 * (size_t)smart_groups() + 4);
    HIPCHK(c, hipMemcpyAsync(d_v, v, sizeof(double) * c->D, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_mu, mu, sizeof(double) * c->D, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), c->stream));
    HIPCHK(c, launch_smart_project(c->dbins, c->dX, c->ldx, c->n, c->D, (int)(cluster - 1), d_v, d_mu, c->d_proj, c->d_vals, d_cnt, c->stream));
    unsigned long long h = 0;
    HIPCHK(c, hipMemcpyAsync(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    *count = (int64_t)h;
    if (values && h > 0) HIPCHK(c, hipMemcpy(values, c->d_vals, sizeof(double) * h, hipMemcpyDeviceToHost));
    return DPMM_OK;
}

int dpmm_smart_kmeans_iter(dpmm_ctx *c, int64_t cluster, double m_lo, double m_hi, double *out4) {
    if (!c || !out4) return DPMM_EINVAL;
    if (int rc = smart_check(c, cluster)) return rc;
    if (!c->d_proj) return fail(c, DPMM_ESTATE, "dpmm_smart_project has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    out4[0] = out4[1] = out4[2] = out4[3] = 0.;
    if (c->n == 0) return DPMM_OK;
    double *d_out = c->d_smart + 4 * (size_t)smart_groups();
    HIPCHK(c, launch_smart_kmeans(c->dbins, c->d_proj, c->n, (int)(cluster - 1), m_lo, m_hi, c->d_smart, d_out, c->stream));
    HIPCHK(c, hipMemcpyAsync(out4, d_out, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    return DPMM_OK;
}

int dpmm_smart_assign(dpmm_ctx *c, int64_t cluster, double m_lo, double m_hi) {
    if (!c) return DPMM_EINVAL;
    if (int rc = smart_check(c, cluster)) return rc;
    if (!c->d_proj) return fail(c, DPMM_ESTATE, "dpmm_smart_project has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n > 0) HIPCHK(c, launch_smart_assign(c->dbins, c->d_proj, c->n, (int)(cluster - 1), m_lo, m_hi, c->stream));
    return DPMM_OK;
}

int dpmm_set_ground_truth(dpmm_ctx *c, const int64_t *gt, int n_gt) {
    if (!c) return DPMM_EINVAL;
    if ((!gt && c->n > 0) || n_gt < 1 || n_gt > 65536) return fail(c, DPMM_EINVAL, "bad ground truth");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_gt) HIPCHK(c, hipMalloc(&c->d_gt, sizeof(int32_t) * (size_t)std::max<int64_t>(c->n, 1)));
    if (c->d_cont) { hipFree(c->d_cont); c->d_cont = nullptr; }
    HIPCHK(c, hipMalloc(&c->d_cont, sizeof(unsigned long long) * (size_t)DPMM_MAX_CLUSTERS * n_gt));
    if (c->n > 0) {
        int64_t *tmp = nullptr;
        HIPCHK(c, hipMalloc(&tmp, sizeof(int64_t) * (size_t)c->n));
        hipError_t e = hipMemcpyAsync(tmp, gt, sizeof(int64_t) * c->n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = launch_i64_to_i32(c->d_gt, tmp, c->n, c->stream);
        if (e == hipSuccess) e = sync_stream(c, c->stream);
        hipFree(tmp);
        if (e != hipSuccess) { c->err = std::string("dpmm_set_ground_truth: ") + hipGetErrorString(e); return DPMM_EHIP; }
    }
    c->n_gt = n_gt;
    return DPMM_OK;
}

int dpmm_contingency(dpmm_ctx *c, int K, int64_t *counts) {
    if (!c || !counts) return DPMM_EINVAL;
    if (!c->d_gt || !c->have_labels) return fail(c, DPMM_ESTATE, "contingency needs labels and dpmm_set_ground_truth");
    if (int rc = check_K(c, K)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = sizeof(unsigned long long) * (size_t)K * c->n_gt;
    if (int rc = ensure_pinned(c, bytes)) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_cont, 0, bytes, c->stream));
    if (c->n > 0) HIPCHK(c, launch_contingency(c->dbins, c->d_gt, c->n, K, c->n_gt, c->d_cont, c->stream));
    HIPCHK(c, launch_copy_bytes(c->h_pin, c->d_cont, bytes, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    memcpy(counts, c->h_pin, bytes);
    return DPMM_OK;
}

int dpmm_sync(dpmm_ctx *c) {
    if (!c) return DPMM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_stream(c, c->stream));
    return DPMM_OK;
}

#ifdef DPMM_STAMPS
// diagnostic builds only: copy the per-wave phase cycle sums of the last NIW sweep
int dpmm_dev_stamps(dpmm_ctx *c, unsigned long long *out, int nwaves) {
    hipStreamSynchronize(c->stream);
    hipMemcpy(out, g_dbg, sizeof(unsigned long long) * 16 * nwaves, hipMemcpyDeviceToHost);
    return c->sweep_grid * 4;
}
#endif

void *dpmm_stream(dpmm_ctx *c) { return c ? (void *)c->stream : nullptr; }

int dpmm_set_option(dpmm_ctx *c, int option, double value) {
    if (!c) return DPMM_EINVAL;
    switch (option) {
        case DPMM_OPT_SCREEN_MARGIN: if (!(value >= 0)) return fail(c, DPMM_EINVAL, "margin < 0"); c->opt_margin = (float)value; return DPMM_OK;
        case DPMM_OPT_TAIL_SCREEN: c->opt_tail = value != 0; return DPMM_OK;
        case DPMM_OPT_PRESCREEN: c->opt_prescreen = value < 0 ? -1 : (value != 0); return DPMM_OK;
        case DPMM_OPT_ORDERED_SWEEP: c->opt_ordered = value != 0; return DPMM_OK;
        case DPMM_OPT_MULT_FORCE_F32:
            if (c->have_points) return fail(c, DPMM_ESTATE, "DPMM_OPT_MULT_FORCE_F32 must be set before the points are uploaded");
            c->opt_force_f32 = value != 0; return DPMM_OK;
        case DPMM_OPT_STATS_ITEMS:
            if (c->Kcap > 0) return fail(c, DPMM_ESTATE, "DPMM_OPT_STATS_ITEMS must be set before the first parameters / K");
            if (value >= 1) c->chunk = (int)std::max<int64_t>((c->prior == DPMM_PRIOR_NIW && c->D <= 64) ? 32 : 256, ((c->n + (int64_t)value - 1) / (int64_t)value + 3) / 4 * 4);
            return DPMM_OK;
        case DPMM_OPT_STATS_GROUPS: c->opt_stats_groups = value > 0 ? (int)value : 0; return DPMM_OK;
        case DPMM_OPT_TRACE_SLOW: c->opt_trace = value != 0; return DPMM_OK;
        case DPMM_OPT_LOGLIK_REF_CONST: c->opt_ref_const = value != 0; return DPMM_OK;
        case DPMM_OPT_WAVE_PRIO: c->opt_prio = value != 0; return DPMM_OK;
        case DPMM_OPT_MULT_DRAWS_AHEAD: c->opt_mult_draws_ahead = value != 0; c->mspec_valid = false; return DPMM_OK;
        case DPMM_OPT_SWEEP_QUEUE_ROUNDS: c->opt_queue_rounds = value < 0 ? -1 : (int)value; return DPMM_OK;
        case DPMM_OPT_BALL_SCREEN: c->opt_ball = value != 0; return DPMM_OK;
        case DPMM_OPT_STATS_DERIVE: c->opt_derive = value != 0; c->cache_force = true; return DPMM_OK;
        case DPMM_OPT_NOISE_AHEAD: c->opt_noise_ahead = value < 0 ? -1 : (value != 0); return DPMM_OK;
        case DPMM_OPT_REF_BRACKET: c->opt_bracket = value != 0; return DPMM_OK;
        case DPMM_OPT_COMM_TIMEOUT_MS: c->comm_timeout_ms = value > 0 ? (int)std::min(value, 2.0e9) : 0; return DPMM_OK;
        case DPMM_OPT_BF16_SCREENS: c->opt_bf16scr = value != 0; return DPMM_OK;
        case DPMM_OPT_MASTER_POLL: c->opt_master_poll = value != 0; return DPMM_OK;
        case DPMM_OPT_PAIR_BALL: c->opt_pair_ball = value != 0; if (!c->opt_pair_ball) c->have_pb = false; return DPMM_OK;      // (ON takes effect with the next parameter set)
        case DPMM_OPT_LEAN_DIRECTION: c->opt_lean_dir = value != 0; c->lean_off = 0; c->lean_backoff = 15; c->lean_ran = false; return DPMM_OK;
        case DPMM_OPT_CHAIN_FUSION: c->opt_chain = value < 0 ? (0x7fffffff & ~(8 | 32)) : (int)value; return DPMM_OK;
        case DPMM_OPT_LEAN_TILES: c->opt_lean = value != 0; c->lean_off = 0; c->lean_backoff = 15; c->lean_ran = false; return DPMM_OK;
        case DPMM_OPT_B3_SUBLABELS: c->opt_b3 = value != 0; if (!c->opt_b3) c->have_b3 = false; return DPMM_OK;      // (switching it ON takes effect with the next parameter set: its images are packed behind the parameters)
        case DPMM_OPT_DIRECTION_SCREEN:
            c->opt_direction = value < 0 ? -1 : (value != 0);
            if (c->opt_direction == 0) { c->sp_ready = false; c->sp_regime = false; }
            return DPMM_OK;
        case DPMM_OPT_ONE_COLLECTIVE: c->opt_one_collective = value < 0 ? -1 : (value != 0); return DPMM_OK;
        case DPMM_OPT_SORT_TILE: {
            const int t = (int)value;
            if (t != SORT_TILE && t != SORT_TILE_SMALL) return fail(c, DPMM_EINVAL, "DPMM_OPT_SORT_TILE: 512 or 2048");
            if (t < c->sort_tile_min) {          // tables for the smaller tile: allocated when asked for (behind everything that still reads the old ones)
                if (c->n > 16000000) return fail(c, DPMM_ELIMIT, "DPMM_OPT_SORT_TILE: 512-point tiles are limited to shards of 16e6 points");
                HIPCHK(c, hipSetDevice(c->device));
                HIPCHK(c, sync_stream(c, c->stream));
                const int nt = (int)((c->n + t - 1) / t);
                const size_t nbmax = 2 * DPMM_MAX_CLUSTERS;
                int32_t *th = nullptr, *tc = nullptr, *tsp = nullptr;
                HIPCHK(c, hipMalloc(&th, sizeof(int32_t) * nbmax * (size_t)std::max(1, nt)));
                if (hipMalloc(&tc, sizeof(int32_t) * nbmax * (size_t)std::max(1, nt)) != hipSuccess) { hipFree(th); return fail(c, DPMM_EHIP, "DPMM_OPT_SORT_TILE: out of device memory for the 512-point tile tables"); }
                if (hipMalloc(&tsp, sizeof(int32_t) * (size_t)STEP_SPEC_MAX_BINS * (size_t)std::max(1, nt)) != hipSuccess) { hipFree(th); hipFree(tc); return fail(c, DPMM_EHIP, "DPMM_OPT_SORT_TILE: out of device memory for the 512-point tile tables"); }
                hipFree(c->sb.tile_hist); hipFree(c->sb.tile_cnt); hipFree(c->sb.tile_spec);
                c->sb.tile_hist = th; c->sb.tile_cnt = tc; c->sb.tile_spec = tsp;
                c->nt_sort = nt; c->sort_tile_min = t;
            }
            c->sb.tile = t; return DPMM_OK;          // (the tile tables are rebuilt by every pass; perm stays a valid order)
        }
        case DPMM_OPT_KERNEL_TIMING:
            c->opt_timing = (int)value & 15;
            if (!(c->opt_timing & 1)) c->have_sweep_ev = false;
            if (!(c->opt_timing & 2)) c->have_stats_ev = false;
            if (!(c->opt_timing & 4)) c->have_comm_ev[0] = c->have_comm_ev[1] = false;
            return DPMM_OK;
        case DPMM_OPT_SWEEP_GRID:
            if (value > 0) c->sweep_grid = (int)std::min<double>(std::min<double>(value, (double)c->sweep_grid_max), (double)std::max<int64_t>(1, c->ntiles));   // scratch is sized for the default
            return DPMM_OK;
        case DPMM_OPT_MULT_NO_U8:
            if (c->have_points) return fail(c, DPMM_ESTATE, "DPMM_OPT_MULT_NO_U8 must be set before the points are uploaded");
            c->opt_no_u8 = value != 0; return DPMM_OK;
        default: return fail(c, DPMM_EINVAL, "unknown option");
    }
}

int dpmm_last_sweep_work(dpmm_ctx *c, uint64_t *out16) {
    if (!c || !out16) return DPMM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_stream(c, c->stream));
    std::vector<unsigned long long> h((size_t)DPMM_WORK_PER_WAVE * (size_t)c->work_waves);
    if (!h.empty()) HIPCHK(c, hipMemcpy(h.data(), c->d_work + DPMM_WORK_SLOTS, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    unsigned long long w7_lo = 0, w7_hi = 0;
    for (size_t w = 0; w < (size_t)c->work_waves; ++w) {
        for (int i = 0; i < 4; ++i) out16[i] += h[DPMM_WORK_PER_WAVE * w + i];
        out16[8] += h[DPMM_WORK_PER_WAVE * w + 4];
        out16[11] += h[DPMM_WORK_PER_WAVE * w + 5];
        out16[13] += h[DPMM_WORK_PER_WAVE * w + 6];
        // slot 7 packs two counters per wave -- low half: direction screens (8 bf16 matrix instructions per 16 clusters + 4 Float32 row sums
        // each); high half: bf16 three-plane sub-cluster evaluations (niw_lean.hip).  The halves are summed SEPARATELY over the waves (a raw
        // 64-bit sum lets the low halves' carry run into the evaluation count after a few thousand launches between two reads: ADVICE r5)
        w7_lo += h[DPMM_WORK_PER_WAVE * w + 7] & 0xFFFFFFFFull; w7_hi += h[DPMM_WORK_PER_WAVE * w + 7] >> 32;
    }
    out16[15] = std::min<unsigned long long>(w7_lo, 0xFFFFFFFFull) | (std::min<unsigned long long>(w7_hi, 0xFFFFFFFFull) << 32);      // (each half saturates at 2^32 - 1)
    // totals of the launches since the previous call (out16[7] of them); the slots start again from zero
    const long long launches = c->work_launches;
    if (!h.empty()) HIPCHK(c, hipMemsetAsync(c->d_work + DPMM_WORK_SLOTS, 0, sizeof(unsigned long long) * h.size(), c->stream));
    c->work_launches = 0; c->work_waves = 0;
    // matrix instructions per unit of work, per WAVE (v_mfma_f32_16x16x4_f32, 2048 flops each)
    int mf_full = 0, mf_scr = 0;
    if (c->prior == DPMM_PRIOR_NIW) {
        const int NB = c->NB, NP = NB * (NB + 1) / 2;
        const int NG = NB <= 4 ? 4 : 2;                       // points per wave / 16 (launch_niw_sweep configurations)
        mf_full = NP * 4 * NG + (NB <= 4 ? NG : 0);           // block pairs x 4 k-steps x NG (+ the ones-MFMA row sums of the direct kernel)
        mf_scr = 4 * NG;                                       // the 16-row screen: one block, 4 k-steps, NG point groups
    }
    out16[4] = (uint64_t)mf_full; out16[5] = (uint64_t)mf_scr; out16[6] = 2048; out16[7] = (uint64_t)launches;
    out16[9] = c->NB == 16 ? 288 : (c->NB == 8 ? 80 : 48); out16[10] = 16384;      // bf16 matrix instructions of a reference bracket per wave tile
    out16[12] = 8; out16[14] = 16;                           // bf16 matrix instructions of a bottom / top screen (+ 4 Float32 row sums per top screen)                          // a reference bracket: 6 fragments x 2 passes x 4 point groups of v_mfma_f32_16x16x32_bf16 (2 * 16 * 16 * 32 flops each)
    return DPMM_OK;
}

// ---- the collective ------------------------------------------------------------------------------------------------------
int dpmm_comm_use_library(const char *path) {
    g_rccl_path = path ? path : "";
    return DPMM_OK;
}

int dpmm_comm_unique_id(void *out128) {
    if (!out128) return DPMM_EINVAL;
    Rccl &r = rccl();
    if (!r.handle || !r.err.empty()) return fail(nullptr, DPMM_ECOMM, r.err.empty() ? "RCCL unavailable" : r.err);
    const int rc = r.GetUniqueId(out128);
    if (rc != 0) return fail(nullptr, DPMM_ECOMM, std::string("ncclGetUniqueId: ") + r.GetErrorString(rc));
    return DPMM_OK;
}

int dpmm_comm_init(dpmm_ctx *c, const void *unique_id128, int rank, int world) {
    if (!c || !unique_id128) return DPMM_EINVAL;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, DPMM_EINVAL, "bad rank / world");
    Rccl &r = rccl();
    if (!r.handle || !r.err.empty()) return fail(c, DPMM_ECOMM, r.err.empty() ? "RCCL unavailable" : r.err);
    HIPCHK(c, hipSetDevice(c->device));
    comm_release(c);
    UidByValue id;
    memcpy(id.internal, unique_id128, 128);
    void *comm = nullptr;
    const int rc = r.CommInitRank(&comm, world, id, rank);
    if (rc != 0) return fail(c, DPMM_ECOMM, std::string("ncclCommInitRank: ") + r.GetErrorString(rc));
    c->comm = comm; c->rank = rank; c->world = world;
    watchdog_start(c);
    return DPMM_OK;
}

int dpmm_comm_init_host(dpmm_ctx *c, int rank, int world, dpmm_host_allreduce_fn fn, void *user) {
    if (!c || !fn) return DPMM_EINVAL;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, DPMM_EINVAL, "bad rank / world");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_stream(c, c->stream));
    comm_release(c);
    c->host_fn = fn; c->host_user = user; c->rank = rank; c->world = world;
    return DPMM_OK;
}

int dpmm_comm_info(dpmm_ctx *c, int64_t *out8) {
    if (!c || !out8) return DPMM_EINVAL;
    out8[0] = c->world; out8[1] = c->rank;
    out8[2] = c->comm ? 1 : (c->host_fn ? 2 : 0);
    out8[3] = c->comm_bytes[0]; out8[4] = c->comm_bytes[1]; out8[5] = c->comm_calls;
    out8[6] = 0;
    out8[7] = c->last_pass_one_collective ? 1 : 0;      // the last per-step pass used ONE collective (DPMM_OPT_ONE_COLLECTIVE)
    return DPMM_OK;
}

int dpmm_last_comm_ms(dpmm_ctx *c, float *counts_ms, float *rows_ms) {
    if (!c) return DPMM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_stream(c, c->stream));
    if (counts_ms) { *counts_ms = 0.f; if (c->have_comm_ev[0]) HIPCHK(c, hipEventElapsedTime(counts_ms, c->ev_comm[0], c->ev_comm[1])); }
    if (rows_ms) { *rows_ms = 0.f; if (c->have_comm_ev[1]) HIPCHK(c, hipEventElapsedTime(rows_ms, c->ev_comm[2], c->ev_comm[3])); }
    return DPMM_OK;
}

int dpmm_comm_destroy(dpmm_ctx *c) {
    if (!c) return DPMM_EINVAL;
    hipSetDevice(c->device);
    if (c->stream && !c->comm_aborted.load()) sync_stream(c, c->stream);
    comm_release(c);
    return DPMM_OK;
}

int dpmm_comm_allgather_host(dpmm_ctx *c, const void *mine, int64_t bytes, void *all) {
    if (!c || !mine || !all || bytes < 0) return DPMM_EINVAL;
    if (!comm_attached(c) || c->world == 1) { memcpy(all, mine, (size_t)bytes); return DPMM_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = ((size_t)bytes + 7) & ~(size_t)7, tot = nb * (size_t)c->world;
    if (!c->comm) {
        // host transport: every rank contributes its piece in a zeroed [world][nb] buffer; the Int64 sum over the ranks is the gather
        std::vector<int64_t> buf(tot / 8, 0);
        memcpy(reinterpret_cast<char *>(buf.data()) + nb * (size_t)c->rank, mine, (size_t)bytes);
        const int hrc = c->host_fn(c->host_user, buf.data(), (int64_t)(tot / 8), 0);
        if (hrc != 0) return fail(c, DPMM_ECOMM, "host all-reduce callback failed (code " + std::to_string(hrc) + ")");
        for (int rk = 0; rk < c->world; ++rk) memcpy((char *)all + (size_t)rk * (size_t)bytes, reinterpret_cast<char *>(buf.data()) + nb * (size_t)rk, (size_t)bytes);
        return DPMM_OK;
    }
    if (int rc = ensure_pinned(c, nb + tot)) return rc;
    char *dbuf = nullptr;
    HIPCHK(c, hipMalloc(&dbuf, tot));
    HIPCHK(c, sync_stream(c, c->stream));
    memcpy(c->h_pin, mine, (size_t)bytes);
    hipError_t e = launch_copy_bytes(dbuf + nb * (size_t)c->rank, c->h_pin, nb, c->stream);
    int rc = DPMM_OK;
    if (e == hipSuccess) {
        Rccl &r = rccl();
        const int nrc = r.AllGather(dbuf + nb * (size_t)c->rank, dbuf, nb, kNcclInt8, c->comm, c->stream);
        if (nrc != 0) rc = fail(c, DPMM_ECOMM, std::string("ncclAllGather: ") + r.GetErrorString(nrc));
    }
    if (rc == DPMM_OK && e == hipSuccess) e = launch_copy_bytes(c->h_pin + nb, dbuf, tot, c->stream);
    if (e == hipSuccess) e = sync_stream(c, c->stream);
    hipFree(dbuf);
    if (e != hipSuccess) { c->err = std::string("dpmm_comm_allgather_host: ") + hipGetErrorString(e); return DPMM_EHIP; }
    if (rc != DPMM_OK) return rc;
    for (int rk = 0; rk < c->world; ++rk) memcpy((char *)all + (size_t)rk * (size_t)bytes, c->h_pin + nb + nb * (size_t)rk, (size_t)bytes);
    return DPMM_OK;
}

// Sub-cluster log-likelihood table of the CURRENT parameters: out[(2k+s) * n_local + i] = loglik of point i under sub-cluster
// s of cluster k + log lr_weights[k][s] -- the two values create_subclusters_labels! (local_clusters_actions.jl:83-95) draws from
// for a point labelled k.  Same arithmetic as the sub-label phase of dpmm_sweep (the sub-cluster matrices are evaluated as
// the cluster-level rows of a temporary 2K-cluster parameter set, full table, no screening).
int dpmm_debug_subloglik(dpmm_ctx *c, float *out) {
    if (!c || !out) return DPMM_EINVAL;
    if (!c->have_points || !c->have_params || c->predictive) return fail(c, DPMM_ESTATE, "debug_subloglik needs points and sweep parameters");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) return DPMM_OK;
    const int K = c->K, K2 = 2 * K;
    if (c->prior == DPMM_PRIOR_NIW && c->have_b3 && c->opt_b3 && c->have_tail && !c->predictive) {      // (the same condition as run_sweep's)
        // the sweeps' sub-label phase runs the three-plane bf16 evaluation (niw_lean.hip): the same device functions, for every point and cluster
        NiwSweepArgs a{};
        a.X = c->dX; a.ldx = c->ldx; a.n = c->n; a.K = K; a.mup = c->d_mup; a.cst = c->d_cst; a.tail = c->d_tail;
        float *tab = nullptr;
        hipError_t e = hipMalloc(&tab, sizeof(float) * (size_t)K2 * (size_t)c->n);
        if (e == hipSuccess) e = launch_niw_b3_debug(a, tab, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out, tab, sizeof(float) * (size_t)K2 * (size_t)c->n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = sync_stream(c, c->stream);
        hipFree(tab);
        if (e != hipSuccess) { c->err = std::string("dpmm_debug_subloglik: ") + hipGetErrorString(e); return DPMM_EHIP; }
        return DPMM_OK;
    }
    if (K2 > DPMM_MAX_CLUSTERS) return fail(c, DPMM_ELIMIT, "debug_subloglik: 2K > DPMM_MAX_CLUSTERS");
    const int64_t stride = c->ntiles * c->tile;
    const bool niw = c->prior == DPMM_PRIOR_NIW;
    // temporary parameter images: row 3j of the temporary set <- row 3k+1+s of the live one (j = 2k+s)
    const size_t NP = niw ? (size_t)c->NB * (c->NB + 1) / 2 : 0, matsz = NP * 256, dp = niw ? (size_t)16 * c->NB : 0;
    float *tRp = nullptr, *tmu = nullptr, *tcst = nullptr, *table = nullptr, *traw = nullptr;
    uint32_t *tL16 = nullptr;
    hipError_t e = hipMalloc(&tcst, sizeof(float) * 3 * K2);
    if (e == hipSuccess) e = hipMalloc(&table, sizeof(float) * (size_t)(niw ? K2 : 3 * K2) * (size_t)stride);
    int rc = DPMM_OK;
    float *sRp = c->d_Rp, *smu = c->d_mup, *scst = c->d_cst, *sraw = c->d_raw;
    const bool s_rp_current = c->rp_current;
    uint32_t *sL16 = c->d_Lp16;
    const int sK = c->K;
    const bool s_tail = c->have_tail, s_prep = c->have_screen_prep;
    if (e == hipSuccess && niw) {
        e = hipMalloc(&tRp, sizeof(float) * 3 * K2 * matsz);
        if (e == hipSuccess) e = hipMalloc(&tmu, sizeof(float) * 3 * K2 * dp);
        for (int j = 0; j < K2 && e == hipSuccess; ++j) {
            const int src = 3 * (j / 2) + 1 + (j % 2);
            e = hipMemcpyAsync(tRp + (size_t)(3 * j) * matsz, c->d_Rp + (size_t)src * matsz, sizeof(float) * matsz, hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(tmu + (size_t)(3 * j) * dp, c->d_mup + (size_t)src * dp, sizeof(float) * dp, hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(tcst + 3 * j, c->d_cst + src, sizeof(float), hipMemcpyDeviceToDevice, c->stream);
        }
        if (e == hipSuccess) { c->d_Rp = tRp; c->d_mup = tmu; c->d_cst = tcst; }
    } else if (e == hipSuccess) {
        // Multinomial: re-pack a raw row image with the sub-cluster rows in the cluster-level positions
        e = hipMalloc(&traw, sizeof(float) * 3 * K2 * (size_t)c->ldx);
        if (e == hipSuccess) e = hipMemsetAsync(traw, 0, sizeof(float) * 3 * K2 * (size_t)c->ldx, c->stream);
        const size_t NT = (size_t)(c->ldx + 15) / 16, NRB = (size_t)(3 * K2 + 15) / 16;
        if (e == hipSuccess) e = hipMalloc(&tRp, sizeof(float) * NRB * NT * 256);
        if (e == hipSuccess) e = hipMalloc(&tL16, sizeof(uint32_t) * std::max(mult_pack_bf16_words(3 * K2, c->ldx), mult_pack_u8_words(3 * K2, (c->D + 127) / 128 * 128)));
        if (e == hipSuccess) e = hipMemsetAsync(tcst, 0, sizeof(float) * 3 * K2, c->stream);
        for (int j = 0; j < K2 && e == hipSuccess; ++j) {
            const int src = 3 * (j / 2) + 1 + (j % 2);
            e = hipMemcpyAsync(traw + (size_t)(3 * j) * c->ldx, c->d_raw + (size_t)src * c->ldx, sizeof(float) * c->ldx, hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(tcst + 3 * j, c->d_cst + src, sizeof(float), hipMemcpyDeviceToDevice, c->stream);
        }
        if (e == hipSuccess) e = launch_mult_pack(traw, tRp, 3 * K2, c->ldx, c->stream);
        if (e == hipSuccess && c->x_u8) e = launch_mult_pack_u8(traw, tL16, 3 * K2, c->ldx, c->ld8, c->stream);
        else if (e == hipSuccess && c->x_bf16_exact) e = launch_mult_pack_bf16(traw, tL16, 3 * K2, c->ldx, c->stream);
        if (e == hipSuccess) { c->d_Rp = tRp; c->d_cst = tcst; c->d_Lp16 = tL16; c->d_raw = traw; c->rp_current = true; }
    }
    if (e == hipSuccess) {
        c->K = K2; c->have_tail = false; c->have_screen_prep = false;
        rc = run_sweep(c, 0, 0, table, stride);
        if (rc == DPMM_OK) {
            const size_t src_pitch = sizeof(float) * stride * (niw ? 1 : 3);
            e = hipMemcpy2DAsync(out, sizeof(float) * c->n, table, src_pitch, sizeof(float) * c->n, (size_t)K2, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = sync_stream(c, c->stream);
        }
    }
    hipStreamSynchronize(c->stream);
    c->d_Rp = sRp; c->d_mup = smu; c->d_cst = scst; c->d_Lp16 = sL16; c->d_raw = sraw; c->rp_current = s_rp_current; c->K = sK; c->have_tail = s_tail; c->have_screen_prep = s_prep;
    hipFree(tRp); hipFree(tmu); hipFree(tcst); hipFree(table); hipFree(traw); hipFree(tL16);
    if (e != hipSuccess) { c->err = std::string("dpmm_debug_subloglik: ") + hipGetErrorString(e); return DPMM_EHIP; }
    return rc;
}

// Diagnostic for the reference bracket of the D <= 64 sweep (niw_sweep.hip, ref_bracket): q_hi[i] = the bracket's certified upper end of
// q_k(x_i) = |R_k (x_i - mu_k)|^2 for the cluster-level factor of `cluster` (1-based) and q[i] = the Float32 evaluation the sweep would
// make in its place -- same device functions and operand images as the sweep.  c_override > 0 replaces the library's rounding constant.
int dpmm_debug_ref_bracket(dpmm_ctx *c, int64_t cluster, float c_override, float *q_hi, float *q) {
    if (!c || !q_hi || !q) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_NIW || c->NB != 4) return fail(c, DPMM_ESTATE, "the reference bracket exists for the NIW prior with D in 33..64 only");
    if (!c->have_points || !c->have_params || c->predictive || !c->have_tail) return fail(c, DPMM_ESTATE, "debug_ref_bracket needs points and sweep parameters with tail records (D % 4 == 0, K > 2)");
    if (cluster < 1 || cluster > c->K) return fail(c, DPMM_EINVAL, "cluster index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) return DPMM_OK;
    float *d = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(float) * 2 * (size_t)c->n));
    NiwSweepArgs a{};
    a.X = c->dX; a.ldx = c->ldx; a.n = c->n; a.K = c->K; a.Rp = c->d_Rp; a.mup = c->d_mup; a.tail = c->d_tail;
    hipError_t e = launch_niw_refb_debug(a, (int)cluster - 1, c_override, d, d + c->n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(q_hi, d, sizeof(float) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(q, d + c->n, sizeof(float) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = sync_stream(c, c->stream);
    hipFree(d);
    if (e != hipSuccess) { c->err = std::string("dpmm_debug_ref_bracket: ") + hipGetErrorString(e); return DPMM_EHIP; }
    return DPMM_OK;
}

int dpmm_debug_pair_ball(dpmm_ctx *c, float *pd, float *sn) {
    if (!c || !pd || !sn) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_NIW || c->NB != 4) return fail(c, DPMM_ESTATE, "the pair-ball table exists for the NIW prior with D in 33..64 only");
    if (!c->have_params || !c->have_pb) return fail(c, DPMM_ESTATE, "no pair-ball table for the parameters on the device (DPMM_OPT_PAIR_BALL, 2 <= K <= 256, tail records)");
    HIPCHK(c, hipSetDevice(c->device));
    const float *t = c->d_tail + niw_pair_ball_offset((size_t)c->K);
    const size_t K = (size_t)c->K;
    HIPCHK(c, hipMemcpyAsync(pd, t, sizeof(float) * K * K, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(sn, t + K * K, sizeof(float) * K, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c, c->stream));
    return DPMM_OK;
}

int dpmm_debug_bracket_big(dpmm_ctx *c, float *aref, uint32_t *tile_flags) {
    if (!c || !aref || !tile_flags) return DPMM_EINVAL;
    if (c->prior != DPMM_PRIOR_NIW || (c->NB != 8 && c->NB != 16)) return fail(c, DPMM_ESTATE, "this bracket exists for the NIW prior with D = 65 .. 256 only");
    if (!c->have_points || !c->have_params || c->predictive || !c->have_refb_big || !c->have_labels)
        return fail(c, DPMM_ESTATE, "debug_bracket_big needs points, labels and sweep parameters with tail records (D % 4 == 0, K > 2, DPMM_OPT_REF_BRACKET on)");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n == 0) return DPMM_OK;
    const size_t nt = (size_t)((c->n + 127) / 128);
    float *d = nullptr; uint32_t *f = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(float) * nt * 128));
    if (hipMalloc(&f, sizeof(uint32_t) * nt) != hipSuccess) { hipFree(d); return fail(c, DPMM_EHIP, "hipMalloc"); }
    NiwSweepArgs a{};
    a.X = c->dX; a.ldx = c->ldx; a.n = c->n; a.K = c->K; a.mup = c->d_mup; a.cst = c->d_cst; a.bins = c->dbins;
    a.order = (c->have_perm && c->opt_ordered) ? c->sb.perm : nullptr; a.order_total = c->sb.perm_total;
    hipError_t e = hipMemsetAsync(d, 0, sizeof(float) * nt * 128, c->stream);
    if (e == hipSuccess) e = launch_niw_bracket_big(c->NB, a, c->d_refb_big, f, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(aref, d, sizeof(float) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tile_flags, f, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = sync_stream(c, c->stream);
    hipFree(d); hipFree(f);
    if (e != hipSuccess) { c->err = std::string("dpmm_debug_bracket_big: ") + hipGetErrorString(e); return DPMM_EHIP; }
    return DPMM_OK;
}

int dpmm_last_sweep_parts_ms(dpmm_ctx *c, float *out3) {
    if (!c || !out3) return DPMM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    out3[0] = out3[1] = out3[2] = 0.f;
    if (!c->have_sweep_ev || c->have_parts == 0) return DPMM_OK;
    HIPCHK(c, hipEventSynchronize(c->ev[1]));
    if (c->have_parts == 2) {
        HIPCHK(c, hipEventElapsedTime(&out3[0], c->ev[0], c->ev_part[0]));
        HIPCHK(c, hipEventElapsedTime(&out3[1], c->ev_part[0], c->ev_part[1]));
    } else HIPCHK(c, hipEventElapsedTime(&out3[1], c->ev[0], c->ev_part[1]));
    HIPCHK(c, hipEventElapsedTime(&out3[2], c->ev_part[1], c->ev[1]));
    return DPMM_OK;
}

int dpmm_last_kernel_ms(dpmm_ctx *c, float *sweep_ms, float *stats_ms) {
    if (!c) return DPMM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    // wait for the closing event of each pair only (not for the stream: the draws launched ahead behind the posteriors are still running when
    // the engine reads the times of the step that just ended, and a stream synchronise here would put them on the host's critical path)
    if (sweep_ms) { *sweep_ms = 0.f; if (c->have_sweep_ev) { HIPCHK(c, hipEventSynchronize(c->ev[1])); HIPCHK(c, hipEventElapsedTime(sweep_ms, c->ev[0], c->ev[1])); } }
    if (stats_ms) { *stats_ms = 0.f; if (c->have_stats_ev) { HIPCHK(c, hipEventSynchronize(c->ev[3])); HIPCHK(c, hipEventElapsedTime(stats_ms, c->ev[2], c->ev[3])); } }
    return DPMM_OK;
}

}  // extern "C"
#pragma GCC visibility pop
