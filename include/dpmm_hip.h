/*
 * dpmm_hip.h -- C ABI of libdpmmhip.so: the MI355X (gfx950) worker path of the
 * DPMMSubClusters.jl restricted-Gibbs sweep.
 *
 * The reference has no FFI; its seam is the set of WORKER functions the master process
 * invokes through Distributed RPC on each worker's `localpart` of the DArrays.  One
 * dpmm_ctx == one reference worker == one GPU holding one contiguous column range of the
 * D x N Float32 data.  Every entry point below names the reference function(s) it stands
 * in for (paths relative to the reference checkout).
 *
 * Conventions
 *   - all functions return 0 on success, a negative DPMM_E* code otherwise;
 *     dpmm_last_error(ctx) returns a human-readable message for the last failure.
 *   - the caller owns every host buffer; the library copies in/out and never keeps host
 *     pointers after the call returns.  The library owns all device memory of the ctx.
 *   - labels / sub-labels cross the boundary as Int64, 1-based (src/ds.jl:54-55).
 *   - cluster parameter arrays are ordered (cluster, left, right) per cluster: row 3k+w,
 *     w = 0 cluster_dist, 1 l_dist, 2 r_dist (src/ds.jl:29-34 thin_cluster_params).
 *   - one ctx is used from one host thread at a time; calls are stream-ordered in call
 *     order (the reference relies on per-worker FIFO task order,
 *     src/local_clusters_actions.jl:65-67,106-108).
 *   - randomness is counter-based (Philox4x32-10): key = seed, counter = (global point
 *     index, epoch, stream).  `epoch` is supplied by the caller and must be unique per
 *     randomised call; results do not depend on how N is sharded over contexts.
 *   - there is NO CPU fallback: every entry point that computes fails with
 *     DPMM_ENODEVICE when no gfx950 device is usable.
 */
#ifndef DPMM_HIP_H
#define DPMM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPMM_ABI_VERSION 3   /* 3 (round 5): dpmm_last_sweep_work fills 16 words (was 8 in version 2), DPMM_OPT_NOISE_AHEAD's -1 = automatic, options 18..27 */

typedef struct dpmm_ctx dpmm_ctx;

enum { DPMM_PRIOR_NIW = 0, DPMM_PRIOR_MULT = 1 };

enum {
    DPMM_OK = 0,
    DPMM_EINVAL = -1,    /* bad argument */
    DPMM_ENODEVICE = -2, /* no usable HIP device */
    DPMM_EHIP = -3,      /* HIP runtime error (message has the hipError string) */
    DPMM_ESTATE = -4,    /* call order violated (e.g. sweep before params) */
    DPMM_ELIMIT = -5,    /* K or D beyond what this build supports */
    DPMM_ECOMM = -6      /* RCCL unavailable or a collective failed (message has the RCCL error string) */
};

/* Behaviour switches (dpmm_set_option).  The library takes nothing from the environment. */
enum {
    DPMM_OPT_SCREEN_MARGIN = 1,   /* NIW sweep: clusters provably more than `value` nats below a point's reference cluster for a whole
                                     wave are skipped (default 50; 0 evaluates every cluster in full) */
    DPMM_OPT_TAIL_SCREEN = 2,     /* 0 / 1: the 4-row tail screen in front of the 16-row MFMA screen (default 1) */
    DPMM_OPT_PRESCREEN = 3,       /* -1 auto (K > 64 or no tail screen) / 0 / 1: triangle-inequality far mask */
    DPMM_OPT_ORDERED_SWEEP = 4,   /* 0 / 1: visit points in the order of the previous statistics pass (default 1) */
    DPMM_OPT_MULT_FORCE_F32 = 5,  /* Multinomial: Float32 matrix-core kernel even for bf16-exact count data (before upload) */
    DPMM_OPT_STATS_ITEMS = 6,     /* work items of the statistics pass (before the first parameters) */
    DPMM_OPT_STATS_GROUPS = 7,    /* workgroups of the NIW statistics kernel (0: default) */
    DPMM_OPT_TRACE_SLOW = 8,      /* log HIP calls that block for more than 5 ms to stderr */
    DPMM_OPT_LOGLIK_REF_CONST = 9,/* dpmm_debug_loglik adds back the reference's -D*D/2*log(2 pi) normaliser (mv_gaussian.jl:24) */
    DPMM_OPT_MULT_NO_U8 = 11,     /* Multinomial: do not keep the byte copy of integer data in [0, 255] (before upload; A/B and tests) */
    DPMM_OPT_SWEEP_GRID = 12,     /* workgroups of the sweep kernels, at most the default (compute units x resident workgroups per unit); experiments */
    DPMM_OPT_SWEEP_QUEUE_ROUNDS = 13, /* D <= 64 NIW sweep: the last rounds of tiles handed out through the queue; -1 (default): rounds / 8, at least 2, none below 4 rounds; 0: static schedule */
    DPMM_OPT_BALL_SCREEN = 14,        /* 1 (default): cluster-per-lane ball test in front of the per-point 4-row tail screen of the NIW sweeps; 0: per-point screens only (same labels) */
    DPMM_OPT_STATS_DERIVE = 16,       /* 1 (default): the per-step statistics pass computes only the SMALLER sub-cluster of every cluster no point entered or
                                         left since its cluster-level statistics were cached, and takes the other one as cache - computed (Float64; the
                                         derived side is the larger one); 0: both sub-clusters of every cluster, every pass */
    DPMM_OPT_NOISE_AHEAD = 17,        /* device master: 1 = the normals of the next parameter draws are generated on a second stream beside the sweep
                                         (measured per step: D = 256 shard 2.16 -> 2.09 ms, D = 128 1.10 -> 1.08, D = 64 shard of 1.25e6 points 0.390 ->
                                         0.383, nothing at N = 1e7); 0 = every draw kernel generates its own: one stream, no cross-stream dependency;
                                         -1 (default) = 1 for D >= 128 or fewer than 4e6 points on this worker, else 0;
                                         same draws either way */
    DPMM_OPT_REF_BRACKET = 18,        /* 1 (default): D in 33..64 sweep, waves whose points all had the same label (D in 65..256: 128-point tiles, in a launch of its own
                                         in front of the sweep): the reference cluster's value is first BRACKETED
                                         (two bf16 matrix passes with a certified rounding bound, ~1/7 of a Float32 evaluation) and evaluated in Float32 only
                                         if some other cluster survives the screens against the bracket's lower end; same labels; 0: always evaluated */
    DPMM_OPT_SORT_TILE = 19,          /* points per sorting wave of the statistics passes: 512 (default below 4e6 points per shard) or 2048 */
    DPMM_OPT_COMM_TIMEOUT_MS = 22,    /* RCCL transport: the longest a host call may block on the ctx stream behind a collective (default 1 800 000 ms = half an hour -- a peer that
                                         pauses on its host looks like a dead one from here, and the reference would wait; 0: for ever).  Only waits on a stream that holds
                                         an enqueued collective are timed.  Past the limit a watchdog aborts the communicator and the blocked call -- on every surviving
                                         rank -- fails with DPMM_ECOMM; later collectives refuse with DPMM_ECOMM until dpmm_comm_release, calls without a collective
                                         (dpmm_get_labels, dpmm_sync) keep working */
    /* 24: was DPMM_OPT_F32_STATS (rounds 4: centred Float32 second moments, +4 % on the headline); removed in round 5 -- the statistics are Float64 throughout, as the reference's (priors/niw.jl:42-51) */
    DPMM_OPT_B3_SUBLABELS = 26,       /* 1 (default): NIW sweep at D in 33..64 with tail records (D % 4 == 0, K > 1): the two sub-cluster quadratic forms of every label through
                                         three-plane bf16 images (v = h + m + l exactly; six of the nine plane products; z = x - mu_k converted once per label,
                                         R_s (x - mu_s) = R_s z + d_s) on v_mfma_f32_16x16x32_bf16 -- Float32-equivalent accuracy, another association of the same products --
                                         in kernels of their own (niw_lean.hip): tiles the cheap screens settle completely in one launch, the rest label phase + sub-label
                                         phase.  0: the Float32 chain inside the sweep kernel.  dpmm_debug_subloglik follows the setting.  Takes effect with the next
                                         parameter set. */
    DPMM_OPT_LEAN_TILES = 27,         /* 1 (default; with DPMM_OPT_B3_SUBLABELS): tiles whose points all had one label and for which the reference bracket, the ball test and the
                                         4-row tail screens exclude every other cluster are finished -- label and sub-labels -- by niw_lean_kernel (its tiles aligned to the
                                         bins of the last sort); what it hands on is finished by one launch of the sweep kernel.  Same labels and sub-labels either way.
                                         0: every tile through the sweep kernel (labels) and niw_sub_kernel (sub-labels).
                                         A sweep that hands on more than 30 % of its tiles switches the lean launch off for 15 sweeps (31, 63, ... up to 1023 while the
                                         retries keep failing); with DPMM_OPT_LEAN_DIRECTION = 0 it also stays off while the direction screen's regime is on (rounds 4-5), and -- on the dpmm_set_params_* path only -- beyond 64 clusters, where the
                                         scalar pre-screen (DPMM_OPT_PRESCREEN) runs inside the sweep kernel.  Parameters drawn on the device (dpmm_niw_master_draw) come without a pre-screen: there
                                         the lean launch runs at any K (tiles aligned to the sort's bins up to 256 bins, 64 consecutive positions beyond; tests/test_gpu_niw.py). */
    DPMM_OPT_MASTER_POLL = 28,        /* 1 (default, round 6): dpmm_step_master_device waits on the posteriors' own records in pinned host memory (every record starts as a
                                         marker no kernel produces) instead of on an event: no barrier packet between the posteriors and the draws launched behind them.
                                         0: the event wait of rounds 3-5.  Same values either way. */
    DPMM_OPT_CHAIN_FUSION = 29,       /* bit mask (default, also for a negative value: 1 | 2 | 4 | 16; round 6): launches of the n-independent chain of a step folded into their neighbours --
                                         1: the sort's bin / item starts inside the scatter launch (no starts_step launch); 2: the three-plane sub-cluster images
                                         written by the launch that packs the parameters (no niw_b3_pack launch); 4: the list of pooled-pair jobs that ride in the posteriors'
                                         launch is read from pinned host memory (no copy launch when the merge gates change); 8: (off by default: measured, no gain) the bad-cluster reset is counted ahead by the per-step
                                         histogram for the clusters that are one-sided in a tile and applied by the scatter launch (no reset_recount launch; K <= 256; from 4e6
                                         points per shard, with bit 32 at any size); 16: the standard
                                         normals of the draws launched ahead are generated by extra workgroups of the posteriors' launch (D <= 128; no kernel on the second stream beside
                                         the sweep, no cross-stream wait in front of the draws).  0: the launches of round 5.  Same values. */
    DPMM_OPT_LEAN_DIRECTION = 30,     /* 1 (default, round 6): while the direction screen's tables exist (DPMM_OPT_DIRECTION_SCREEN: overlapping clusters) niw_lean_kernel runs the
                                         screen itself -- on plane h of z0 = x - mu_k0, which it holds -- and settles the tiles it clears; the launch behind it gets a
                                         direction-screen instantiation for the spans handed on.  0: no lean launch in that regime (rounds 4-5).  Same labels and sub-labels. */
    DPMM_OPT_PAIR_BALL = 31,          /* 1 (default, round 6): D in 33..64 NIW sweep, 2 <= K <= 256: with every parameter set the library tabulates (workgroups of a launch that runs anyway), for all pairs
                                         (k, j), a certified lower bound of |R_j (mu_k - mu_j)| -- the distance of cluster k's mean from cluster j's in j's own
                                         metric -- and per cluster an upper bound of |R_j|_2; niw_lean_kernel then excludes cluster j for a whole tile of points
                                         within r of mu_k0 when cst_j - (D_k0,j - |R_j|_2 r)^2 / 2 is below the tile's lowest threshold: the ball test in all D
                                         features, one comparison per cluster and tile, in front of the 4-feature one.  0: without it.  Same labels and sub-labels. */
    DPMM_OPT_MULT_DRAWS_AHEAD = 25,   /* 1 (default): Multinomial device master: dpmm_step_stats launches the NEXT Dirichlet draws and their hand-over images
                                       * behind the statistics (the epoch after the last dpmm_mult_master_draw, the same K and outlier flag), into a second set of
                                       * buffers, and returns when the rows are on the host -- the draws run while the caller decides splits and merges.
                                       * dpmm_mult_master_draw takes them when its arguments are the ones guessed and the rows have not changed; else it draws
                                       * as before.  0: always draw inside dpmm_mult_master_draw (same values either way) */
    DPMM_OPT_DIRECTION_SCREEN = 23,   /* D in 33..64 NIW sweep, K <= 64: a tile that keeps six or more candidate clusters behind the 4-row tests puts ALL of them
                                         through one bound each -- along the direction u = R_k (mu_k0 - mu_k) / b that separates cluster k from the wave's reference
                                         cluster k0: q_k(x) >= (w . (x - mu_k0) + b)^2, the K dot products of a point from one bf16 matrix product (16-32 matrix
                                         instructions per tile instead of 8-24 per CLUSTER) -- before the 16-row screens; needs one small kernel per parameter set
                                         (K^2 direction vectors).  It only removes candidates the Float32 evaluation would have excluded: same labels.
                                         -1 (default): on while the previous sweep's tiles kept 8 or more candidates on average (off again below 4, or when it removes
                                         less than a quarter of what it is given);
                                         0: never; 1: always */
    DPMM_OPT_BF16_SCREENS = 21,       /* 1 (default): D in 33..64 NIW sweep: a bf16 lower bound of the last block row's part of the quadratic form (8 matrix
                                         instructions) in front of every Float32 16-row screen (16), and of the first block row's part (16 + 4) in front of every
                                         survivor's evaluation; they only skip Float32 tests that would have excluded the cluster too: same labels; 0: off */
    DPMM_OPT_ONE_COLLECTIVE = 20,     /* -1 (default): automatic = 1 while a packed row has at most 4096 doubles (D <= 88), else 0.  1: NIW per-step pass (dpmm_step_stats*) with a communicator attached: ONE all-reduce of 3K packed rows -- the 2K rows
                                         of the labels as swept + K re-drawn left rows of the clusters each shard reset speculatively (those with exactly one
                                         empty sub-cluster on the shard); the bad-cluster verdict comes out of the reduced rows' N column, a shard's candidate
                                         that is not bad gets its sub-labels back.  0: occupancy all-reduce -> reset -> statistics -> row all-reduce.  Same chain. */
    DPMM_OPT_KERNEL_TIMING = 15,      /* bit mask: 1 = HIP events around the sweep kernel, 2 = around the statistics pass (dpmm_last_kernel_ms), 4 = around the
                                         all-reduces (dpmm_last_comm_ms), 8 (with 1) = between the up to three launches of a D in 33..64 sweep (dpmm_last_sweep_parts_ms);
                                         0 (default): none -- every event is a barrier packet between two kernels, ~5 us each */
    DPMM_OPT_WAVE_PRIO = 10       /* 0 / 1: NIW sweep (D <= 64) lowers a wave's issue priority while it streams matrix instructions and
                                     raises it in its scalar / VALU phases (default 1) */
};

#define DPMM_MAX_CLUSTERS 1024
#define DPMM_MAX_DIM_NIW 256

int dpmm_abi_version(void);

/* Worker construction: the shard [first_index, first_index + n_local) of the N points.
 * Replaces: the worker-side state created by `distribute` in init_model_from_data
 * (src/dp-parallel-sampling.jl:36-53) -- localpart(points), localpart(labels),
 * localpart(labels_subcluster) -- and Random.seed!(seed) on the worker (:37-39). */
int dpmm_create(dpmm_ctx **ctx, int prior_kind, int D, int64_t n_local, int64_t first_index,
                int device, uint64_t seed);
int dpmm_destroy(dpmm_ctx *ctx);
const char *dpmm_last_error(const dpmm_ctx *ctx); /* ctx may be NULL: last create error */

/* Points of this shard: X is D x n_local Float32, column-major with leading dimension ldx
 * (point i = X + i*ldx), host memory.  Replaces distribute(all_data) (dp-parallel-sampling.jl:42-44).
 * _device: same, but X already lives in device memory of ctx's device (zero-copy hand-over
 * from a producer on the GPU; copied into the ctx-owned layout on the ctx stream). */
int dpmm_upload_points(dpmm_ctx *ctx, const float *X, int64_t ldx);
int dpmm_upload_points_device(dpmm_ctx *ctx, const float *dX, int64_t ldx);
/* .npy ingestion (next row of the scope table): `rows` is the Samples x Dimensions array of a .npy file as the
 * reference's advanced mode reads it (load_data, src/utils.jl:5-14: npzread, NaN -> 0, transpose; then Float32.(...) in
 * init_model, src/dp-parallel-sampling.jl:19-25) -- n_local rows of D elements, Float32 (is_f64 = 0) or Float64
 * (is_f64 = 1), row r at rows + r*ld elements, host memory.  Conversion to Float32, the NaN -> 0 replacement (when
 * nan_to_zero != 0) and the re-layout happen on the device. */
int dpmm_upload_points_npy(dpmm_ctx *ctx, const void *rows, int is_f64, int64_t ld, int nan_to_zero);

/* labels = rand(1:init_clusters), sub-labels = rand(1:2) (dp-parallel-sampling.jl:49-50).
 * _from: labels = first_label - 1 + rand(1:init_clusters) -- `.+ 1` when cluster 1 is the outlier component (:49). */
int dpmm_init_labels(dpmm_ctx *ctx, int init_clusters, uint32_t epoch);
int dpmm_init_labels_from(dpmm_ctx *ctx, int init_clusters, int first_label, uint32_t epoch);
int dpmm_set_option(dpmm_ctx *ctx, int option, double value);
/* Resume / tests: overwrite or read back this shard's labels (Array(group.labels),
 * dp-parallel-sampling.jl:218,276,371).  Either pointer may be NULL. */
int dpmm_set_labels(dpmm_ctx *ctx, const int64_t *labels, const int64_t *sub_labels);
int dpmm_get_labels(dpmm_ctx *ctx, int64_t *labels, int64_t *sub_labels);

/* Cluster parameters for the next sweep.  Replaces broadcast_cluster_params /
 * set_global_data (local_clusters_actions.jl:518-549): the fields of thin_cluster_params the
 * workers actually read -- mv_gaussian mu, invSigma, logdetSigma (distributions/mv_gaussian.jl:12-18),
 * lr_weights, and the mixture weights.
 *   mu        [3K][D]        inv_sigma [3K][D*D] (symmetric; row/column-major agree)
 *   logdet    [3K]           lr_weights [K][2]      weights [K]
 * dpmm_set_params_niw factorises inv_sigma = R'R on the host (compat path, O(K D^3));
 * dpmm_set_params_niw_chol takes the upper-triangular factor R directly
 * (row-major [3K][D][D], entries below the diagonal ignored): the quadratic form is
 * evaluated as ||R (x - mu)||^2. */
int dpmm_set_params_niw(dpmm_ctx *ctx, int K, const float *mu, const float *inv_sigma, const float *logdet,
                        const float *lr_weights, const float *weights);
int dpmm_set_params_niw_chol(dpmm_ctx *ctx, int K, const float *mu, const float *R, const float *logdet,
                             const float *lr_weights, const float *weights);
/* multinomial_dist alpha = log-probabilities (distributions/multinomial_dist.jl:8-10): logp [3K][D] */
int dpmm_set_params_mult(dpmm_ctx *ctx, int K, const float *logp, const float *lr_weights, const float *weights);

/* The per-sweep path of the same hand-over, without copies: dpmm_params_staging returns pointers into pinned, GPU-addressable
 * host memory owned by the ctx, sized for `slots` clusters; the master writes its parameter draws there IN PLACE and
 * dpmm_commit_params(K) packs them for the kernels (which read the staging buffer directly; no synchronisation).
 *   NIW : mu [3 slots][D], mat = R [3 slots][D(D+1)/2] -- the upper-triangular factor PACKED row by row (row r holds
 *         columns r..D-1 at offset r*D - r*(r-1)/2): half the bytes of the square cross the host link --, logdet [3 slots]
 *   MULT: mu = NULL, mat = logp [3 slots][D], logdet = NULL
 *   lr_weights [K][2], weights [K] and slot_of_cluster [K] in cluster order: cluster k's three rows are rows
 *   3*slot_of_cluster[k] + w of mu / mat / logdet -- a cluster keeps its rows in place for life, removing clusters only edits the map.
 * Contents survive a re-allocation (a later call with more slots).  The staging may be written again once a blocking call
 * of the ctx (dpmm_step_stats, dpmm_suffstats_*, dpmm_sync) has returned. */
int dpmm_params_staging(dpmm_ctx *ctx, int slots, float **mu, float **mat, float **logdet, float **lr_weights, float **weights,
                        int32_t **slot_of_cluster);
int dpmm_commit_params(dpmm_ctx *ctx, int K);

/* Declares the number of clusters K that labels may reference from now on WITHOUT uploading
 * parameters: the master resizes group.local_clusters in check_and_split!
 * (local_clusters_actions.jl:361) and shrinks it in remove_empty_clusters! (:470); reference
 * workers learn K implicitly from length(clusters_vector) (:152,:174).  Needed before a
 * statistics pass that follows a split or a removal.  Parameters must be set again before
 * the next dpmm_sweep. */
int dpmm_set_num_clusters(dpmm_ctx *ctx, int K);
int dpmm_num_clusters(const dpmm_ctx *ctx);   /* the K last declared (0 before the first parameters) */
/* NUMA node of the host the ctx's GPU hangs off (sysfs numa_node of its PCI function), -1 when unknown: the master's threads
 * read 1-17 MB of device-written rows and stage as much for the device per sweep -- on a two-socket host they belong on that node. */
int dpmm_numa_node(dpmm_ctx *ctx);

/* One label + sub-label sampling pass over the shard.
 * Replaces sample_labels_worker! (local_clusters_actions.jl:112-134; log_likelihood!
 * mv_gaussian.jl:21-25 / multinomial_dist.jl:13-15; sample_log_cat_array! utils.jl:19-31)
 * followed by sample_sub_clusters_worker! / create_subclusters_labels! (:70-95).
 * final_argmax != 0 selects argmax for the labels (`final`/hard_clustering, :129-130);
 * sub-labels are always sampled. Asynchronous (stream-ordered). */
int dpmm_sweep(dpmm_ctx *ctx, uint32_t epoch, int final_argmax);

/* Sufficient statistics of this shard.
 * Replaces create_suff_stats_dict_worker (local_clusters_actions.jl:149-169) with
 * create_sufficient_statistics (priors/niw.jl:42-51, priors/multinomial_prior.jl:27-32).
 * cluster_idx: 1-based cluster ids to compute (NULL => all K, as `indices == nothing`).
 *
 * The packed form is what crosses GPUs (one all-reduce(sum) replaces the two-level
 * reduce of create_suff_stats_dict_node_leader / update_suff_stats_posterior!,
 * :171-254, aggregate_suff_stats niw.jl:64-66 / multinomial_prior.jl:41-43):
 *   Float64 [2K][dpmm_packed_stride(ctx)] ; row 2k+s (s=0 left/sub==1, s=1 right/sub==2) =
 *   { N, sum[0..D-1], S lower triangle row-major (a>=b: S[a][b]) }   (NIW)
 *   { N, sum[0..D-1] }                                               (Multinomial)
 * Rows of clusters not listed in cluster_idx are zero.  Cluster-level statistics are
 * left + right (the reference recomputes them; equal up to Float64 rounding).
 * dpmm_suffstats_packed_device writes the packed rows into caller-provided DEVICE memory
 * (stream-ordered; use dpmm_sync before reading from another stream);
 * dpmm_suffstats_packed copies them to host memory (synchronous). */
int64_t dpmm_packed_stride(const dpmm_ctx *ctx);
int dpmm_suffstats_packed_device(dpmm_ctx *ctx, const int64_t *cluster_idx, int n_idx, double *d_out);
int dpmm_suffstats_packed(dpmm_ctx *ctx, const int64_t *cluster_idx, int n_idx, double *out);
/* Same statistics, handed over in place: *packed points into pinned host memory of the ctx (valid until its next call).
 * With a communicator attached (dpmm_comm_init) the rows are the SUM over all ranks -- the all-reduce runs on the ctx
 * stream between the statistics kernels and the copy (dpmm_suffstats_packed does the same; _device stays local). */
int dpmm_suffstats_host(dpmm_ctx *ctx, const int64_t *cluster_idx, int n_idx, const double **packed);
/* Steps 5 + 6 of group_step (local_clusters_actions.jl:665-666) in ONE device pass without a host round trip in between:
 * sub-cluster occupancies (summed over the ranks) -> clusters with an empty sub-cluster are flagged and the sub-labels of
 * their points re-drawn with `reset_epoch` (reset_bad_clusters!, :501-516) -> statistics of all K clusters over the final
 * labelling (summed over the ranks).  *packed as dpmm_suffstats_host, *bad [K] flags; blocks until both are in host memory.
 * With DPMM_OPT_STATS_DERIVE (default) the pass accumulates only the SMALLER sub-cluster of every cluster whose membership did not
 * change since its cluster-level row was cached (the histogram tracks every point's label between passes) and returns the other one
 * as cache - accumulated: the same Float64 sums as create_sufficient_statistics (priors/niw.jl:42-51) in another association
 * (N exact, the rest to ~1e-16 relative); dpmm_suffstats_* always accumulate everything they are asked for. */
int dpmm_step_stats(dpmm_ctx *ctx, uint32_t reset_epoch, const double **packed, const uint8_t **bad);
/* Expand packed rows (after any cross-GPU sum) into the reference's thin_suff_stats shape
 * (src/ds.jl:37-41), order (cluster, left, right): N [K][3], sum [K][3][D], S [K][3][D][D]
 * (S symmetric; NULL for Multinomial).  Pure host code. */
int dpmm_unpack_suffstats(const dpmm_ctx *ctx, int K, const double *packed, double *N, double *sum, double *S);

/* Relabel operations (integer bookkeeping, exact).
 * dpmm_split:  split_cluster_local_worker! (local_clusters_actions.jl:265-278)
 * dpmm_merge:  merge_clusters_worker! (:293-304)
 * dpmm_remove_empty: remove_empty_clusters_worker! (:446-455), pts_count [K]
 * dpmm_reset_sublabels: reset_bad_clusters_worker! (:481-488); idx == NULL => every point
 *                       (split_first_cluster_worker!, :257-261)
 * idx / new_idx are 1-based cluster ids, processed pair by pair in order. */
int dpmm_split(dpmm_ctx *ctx, const int64_t *idx, const int64_t *new_idx, int n, uint32_t epoch);
int dpmm_merge(dpmm_ctx *ctx, const int64_t *idx, const int64_t *new_idx, int n);
int dpmm_remove_empty(dpmm_ctx *ctx, const int64_t *pts_count, int K);
int dpmm_reset_sublabels(dpmm_ctx *ctx, const int64_t *idx, int n, uint32_t epoch);


/* Prediction for the points held by the ctx (next row of the scope table: predict / predict_points,
 * src/dp-parallel-sampling.jl:532-537, src/local_clusters_actions.jl:23-40, with posterior_predictive!
 * priors/niw.jl:68-76 and priors/multinomial_prior.jl:45-48).
 *   NIW: cluster k's posterior predictive is MvTDist(df_k, m_k, Sigma_k); pass m [K][D], the upper-triangular
 *        factor R [K][D][D] of Sigma_k^-1 (= R'R), logdet Sigma_k [K], df [K] and the mixture weights [K].
 *   Multinomial: logp [K][D] = log(alpha'/sum(alpha')), weights [K].
 * dpmm_predict then writes parr[k][i] = log predictive density of point i under cluster k + log w_k (Float32,
 * [K][n_local]); the caller takes the row-wise argmax / normalised exponentials as predict_points does.
 * These calls replace the sweep parameters of the ctx: set them again before the next dpmm_sweep. */
int dpmm_set_predictive_niw(dpmm_ctx *ctx, int K, const float *m, const float *R, const float *logdet,
                            const float *df, const float *weights);
int dpmm_set_predictive_mult(dpmm_ctx *ctx, int K, const float *logp, const float *weights);
int dpmm_predict(dpmm_ctx *ctx, float *parr);
/* The rest of predict_points (src/local_clusters_actions.jl:33-40) on the device as well: labels[i] = row-wise argmax (Int64,
 * 1-based; the first NaN wins as in Julia), probs[i*K + k] = exp(parr - rowmax) / rowsum with NaN -> -Inf (n_local x K,
 * row-major; may be NULL).  Same table as dpmm_predict; only the labels and the normalised matrix cross the bus. */
int dpmm_predict_points(dpmm_ctx *ctx, int64_t *labels, float *probs);

/* Sub-cluster occupancy of the shard after a sweep: counts[2k] = #{label == k+1, sub == 1}, counts[2k+1] = #{..., sub == 2}
 * (Int64, [2K], summable across shards).  These are the N fields of the l / r statistics the reference reads in
 * reset_bad_clusters! (src/local_clusters_actions.jl:501-516) to find clusters with an empty sub-cluster.  Returning
 * them BEFORE the statistics pass lets the host reset those sub-labels first (dpmm_reset_sublabels) and run ONE
 * dpmm_suffstats_packed over the final labelling instead of a full pass plus a subset pass: the reference's state
 * after its steps 5-6 (:665-666) is reproduced exactly, because a sub-label reset leaves cluster-level statistics and
 * every other cluster untouched. */
int dpmm_bin_counts(dpmm_ctx *ctx, int64_t *counts);

/* Smart splits (next row of the scope table; opt-in `smart_splits`, Gaussian prior only): the worker halves of
 * smart_cluster_init! (src/local_clusters_actions.jl:555-627).  The master computes the direction v and the centre mu
 * from the cluster's statistics (:557-568) and drives at most max_split_iter 1-D 2-means steps (:593-623).
 *   dpmm_smart_project      tranform_points_worker! (:643-653): t_i = v.(x_i - mu) (Float64) for the points of `cluster`
 *                           (1-based); `values` (capacity n_local, may be NULL) receives them in arbitrary order so
 *                           that the caller can take the percentiles (:650), `count` their number.
 *   dpmm_smart_kmeans_iter  kmeans_iter_worker! (:635-641): out4 = {sum of t on the m_lo side, count, sum on the m_hi
 *                           side, count}; side 1 iff |t - m_lo| < |t - m_hi|.  Summable across shards.
 *   dpmm_smart_assign       set_smart_labels_in_worker! (:629-633): sub-label := side. */
int dpmm_smart_project(dpmm_ctx *ctx, int64_t cluster, const double *v, const double *mu, double *values, int64_t *count);
int dpmm_smart_kmeans_iter(dpmm_ctx *ctx, int64_t cluster, double m_lo, double m_hi, double *out4);
int dpmm_smart_assign(dpmm_ctx *ctx, int64_t cluster, double m_lo, double m_hi);

/* On-device evaluation (next row of the scope table): with a ground truth the reference gathers all N labels to
 * the master EVERY iteration to compute NMI / VI (src/dp-parallel-sampling.jl:370-377).  Here the ground truth of
 * the shard is uploaded once (Int64, any integer ids in [0, n_gt)), and each call returns only the
 * K x n_gt contingency table counts[k][g] = #{i : label_i == k+1, gt_i == g} (Int64, summable across shards). */
int dpmm_set_ground_truth(dpmm_ctx *ctx, const int64_t *gt, int n_gt);
int dpmm_contingency(dpmm_ctx *ctx, int K, int64_t *counts);

/* Block until all queued work of the ctx has completed. */
int dpmm_sync(dpmm_ctx *ctx);
/* The HIP stream (hipStream_t) all work of this ctx is queued on, for callers that
 * want to time with HIP events or order other work against it. */
void *dpmm_stream(dpmm_ctx *ctx);

/* The one exchange of the sweep, inside the library: RCCL all-reduce(sum) of the packed statistics (and of the Int64
 * sub-cluster occupancies) on the ctx stream.  Replaces create_suff_stats_dict_node_leader / update_suff_stats_posterior!'s
 * reduce (local_clusters_actions.jl:171-254) and aggregate_suff_stats (priors/niw.jl:64-66, multinomial_prior.jl:41-43).
 *   dpmm_comm_unique_id   rank 0 creates the 128-byte RCCL id and ships it to the other ranks by any means
 *   dpmm_comm_init        every rank (one process per GPU) joins; collective
 *   dpmm_comm_init_host   the same attachment with the caller's OWN transport instead of RCCL: `fn(user, buf, count, is_f64)` sums `count`
 *                         Float64 (is_f64 = 1) or Int64 (0) values of the HOST buffer `buf` over the ranks in place and returns 0.  The library
 *                         stages the device buffer through pinned memory around the call (one blocking round trip per all-reduce).  This is
 *                         what the reference's own Distributed transport does (fetch of host Dicts, local_clusters_actions.jl:231); it exists
 *                         for hosts without a peer-to-peer fabric between the ranks' GPUs (and lets several ranks share ONE GPU in tests)
 * After either, dpmm_step_stats / dpmm_suffstats_host / dpmm_suffstats_packed return statistics summed over all ranks.
 * dpmm_comm_allgather_host gathers `bytes` of host data from every rank (all [world][bytes]); collective.
 * dpmm_comm_info: out8 = {world, rank, transport (0 none, 1 RCCL, 2 host function), bytes of the last occupancy all-reduce, bytes of the
 * last packed-row all-reduce, all-reduces since the attachment, 0, 1 if the last per-step pass used one collective (DPMM_OPT_ONE_COLLECTIVE)}.  dpmm_last_comm_ms: HIP-event time of the last all-reduce of each
 * kind on the ctx stream (0 if none; synchronises the stream). */
typedef int (*dpmm_host_allreduce_fn)(void *user, void *buf, int64_t count, int is_f64);

/* RCCL is bound at run time (dlopen): a copy already mapped into the process wins, then the soname, then /opt/rocm/lib.
 * dpmm_comm_use_library names the file to use instead (before the first dpmm_comm_* call) -- a host that also runs
 * torch.distributed passes torch's own librccl.so so that the process holds ONE copy. */
int dpmm_comm_use_library(const char *path);
int dpmm_comm_unique_id(void *out128);
int dpmm_comm_init(dpmm_ctx *ctx, const void *unique_id128, int rank, int world);
int dpmm_comm_init_host(dpmm_ctx *ctx, int rank, int world, dpmm_host_allreduce_fn fn, void *user);
int dpmm_comm_info(dpmm_ctx *ctx, int64_t *out8);
int dpmm_comm_destroy(dpmm_ctx *ctx);
int dpmm_comm_allgather_host(dpmm_ctx *ctx, const void *mine, int64_t bytes, void *all);

#ifdef __cplusplus
}
#endif
#endif /* DPMM_HIP_H */
