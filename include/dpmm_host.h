/*
 * dpmm_host.h -- C ABI of libdpmmhost.so: the MASTER half of the DPMMSubClusters.jl restricted-Gibbs sweep.
 *
 * include/dpmm_hip.h replaces what a reference WORKER does on its shard; this header replaces what the reference's
 * master process does between the workers' calls -- the body of group_step (src/local_clusters_actions.jl:658-673):
 *
 *     sample_clusters!                src/local_clusters_actions.jl:417-437, src/shared_actions.jl:41-66
 *     broadcast_cluster_params        src/local_clusters_actions.jl:518-549      (-> worker.commit_params)
 *     sample_labels! / sample_sub_clusters!   :98-109, :64-68                     (-> worker.sweep)
 *     update_suff_stats_posterior!    src/local_clusters_actions.jl:206-254      (-> worker.step_stats / worker.stats)
 *     reset_bad_clusters!             :501-516
 *     check_and_split!                :345-382  (should_split_local! :318-343, split_cluster_local! :280-291)
 *     check_and_merge!                :385-413  (should_merge! shared_actions.jl:21-38, merge_clusters! :308-315)
 *     remove_empty_clusters!          :457-471
 *     init_first_clusters!            src/dp-parallel-sampling.jl:62-78
 *     calculate_posterior             src/dp-parallel-sampling.jl:458-470
 *
 * north_star keeps "posterior cluster-parameter draws and split/merge Metropolis steps on the host": they run here, in
 * native code (threaded over the 3K distributions of a sweep), so that one call -- dpmmh_group_step -- is one sweep and no
 * interpreter sits between the kernels.  A Julia host `ccall`s this header exactly as the Python host does.
 *
 * The model drives a WORKER through a table of C function pointers (dpmmh_worker).  For the GPU path the table is filled
 * with the addresses of the libdpmmhip.so entry points named next to each member (include/dpmm_hip.h); tests fill it
 * with stand-ins.  The model never touches points or labels itself.
 *
 * Conventions: 0 = success, negative = failure (dpmmh_model_last_error has the text); cluster ids crossing this boundary
 * are 1-based as in the reference; "row" 3k+w means distribution w (0 cluster, 1 left, 2 right) of the k-th live cluster.
 * Randomness: Philox4x32-10 keyed by (seed; id, epoch, stream) -- every rank of a multi-GPU run takes identical decisions
 * from identical all-reduced statistics.
 */
#ifndef DPMM_HOST_H
#define DPMM_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPMMH_ABI_VERSION 5

typedef struct dpmmh_model dpmmh_model;

enum { DPMMH_PRIOR_NIW = 0, DPMMH_PRIOR_MULT = 1 };

/* The worker interface (one per process = one GPU shard).  All functions return 0 on success. */
typedef struct dpmmh_worker {
    void *ctx;
    int rank, world;
    /* Persistent host staging for the cluster parameters, sized for `slots` clusters (3 rows each).  Rows are indexed by
     * SLOT (a cluster keeps its slot for life; slot_of_cluster maps the k-th live cluster to its slot):
     *   NIW : mu [3*slots][D], mat = R [3*slots][D(D+1)/2] (upper-triangular factor of Sigma^-1, packed row by row:
     *         row r holds columns r..D-1 at offset r*D - r*(r-1)/2), logdet [3*slots]
     *   MULT: mu = NULL, mat = logp [3*slots][D], logdet = NULL
     *   lr [K][2] and w [K] in cluster order; slot_of_cluster [K].
     * Pointers stay valid until the next call with a larger `slots` (contents are preserved).  dpmm_params_staging */
    int (*params_staging)(void *ctx, int slots, float **mu, float **mat, float **logdet, float **lr, float **w,
                          int32_t **slot_of_cluster);
    int (*commit_params)(void *ctx, int K);                               /* dpmm_commit_params   */
    int (*set_num_clusters)(void *ctx, int K);                            /* dpmm_set_num_clusters */
    int (*sweep)(void *ctx, uint32_t epoch, int final_argmax);            /* dpmm_sweep           */
    /* Steps 5+6 of group_step in one pass: sub-cluster occupancies -> clusters with an empty sub-cluster get their
     * sub-labels re-drawn (reset_bad_clusters_worker!) -> statistics of all K clusters, summed over all ranks.
     * *packed: [2K][stride] rows {N, sum, lower triangle of S}; *bad: [K] flags.  Blocks until the data is in host memory;
     * the pointers are valid until the next worker call.  dpmm_step_stats */
    int (*step_stats)(void *ctx, uint32_t reset_epoch, const double **packed, const uint8_t **bad);
    int (*stats)(void *ctx, const int64_t *cluster_idx, int n_idx, const double **packed);   /* dpmm_suffstats_host */
    int (*split)(void *ctx, const int64_t *idx, const int64_t *new_idx, int n, uint32_t epoch);   /* dpmm_split */
    int (*merge)(void *ctx, const int64_t *idx, const int64_t *new_idx, int n);                   /* dpmm_merge */
    int (*remove_empty)(void *ctx, const int64_t *pts_count, int K);                              /* dpmm_remove_empty */
    int (*reset_sublabels)(void *ctx, const int64_t *idx, int n, uint32_t epoch);                 /* dpmm_reset_sublabels */
    /* labels = first_label - 1 + rand(1:init_clusters), sub-labels = rand(1:2)  (first_label = 2 with an outlier component) */
    int (*init_labels)(void *ctx, int init_clusters, int first_label, uint32_t epoch);            /* dpmm_init_labels_from */
    /* exchange of small host buffers among the ranks (world > 1 only): all[r*bytes ..] = rank r's `mine`.  dpmm_comm_allgather_host */
    int (*allgather)(void *ctx, const void *mine, int64_t bytes, void *all);
    const char *(*last_error)(void *ctx);                                                         /* dpmm_last_error */
    /* OPTIONAL group (all or none; NULL when the worker has no device master): the dense per-distribution maths of the NIW
     * master on the worker's device -- see dpmm_niw_master_* / dpmm_step_stats_device / dpmm_suffstats_device in dpmm_hip_master.h.
     * The engine uses it when DPMMH_OPT_DEVICE_MASTER allows (default: D >= 64) and no outlier prior is set; everything it
     * cannot do there (merge proposals, state access, a restored state) falls back to the host path through niw_rows. */
    int (*niw_master_setup)(void *ctx, double kappa, double nu, const double *m, const double *psi);
    int (*step_stats_device)(void *ctx, uint32_t reset_epoch, const uint8_t **bad);
    int (*step_master_device)(void *ctx, uint32_t reset_epoch, const int32_t *slots, uint32_t draw_epoch, const uint8_t **bad, const double **small);
    int (*stats_device)(void *ctx, const int64_t *cluster_idx, int n_idx);
    int (*niw_posterior)(void *ctx, const int64_t *clusters, const int32_t *slots, int n, const double **small);
    int (*niw_draw)(void *ctx, uint32_t epoch, int K, const int32_t *slot_of_cluster, const float *lr, const float *w);
    int (*niw_pairs)(void *ctx, const int32_t *slots_i, const int32_t *slots_j, int n, const double **small);
    int (*niw_pairs_ahead)(void *ctx, const int32_t *slots_i, const int32_t *slots_j, int n);      /* optional on its own (may be NULL): dpmm_niw_master_pairs_ahead */
    int (*niw_put_rows)(void *ctx, const double *rows, int K);
    int (*niw_rows)(void *ctx, const int32_t *slots, int n, double *out);
    int (*niw_draws)(void *ctx, int K, float *mu, float *R, float *logdet);
    /* OPTIONAL group (all or none): the Multinomial master's parameter draws on the worker's device -- dpmm_mult_master_* in dpmm_hip_master.h.
     * Used when DPMMH_OPT_DEVICE_MASTER allows (default: D >= 128) and the worker's last statistics pass holds the current rows of all K
     * clusters (no split, merge or removal since); otherwise the engine draws on the host as before. */
    int (*mult_master_setup)(void *ctx, const float *alpha, const float *alpha_outlier);
    int (*mult_draw)(void *ctx, uint32_t epoch, int K, int outlier_first, const float *lr, const float *w);
    int (*mult_draws)(void *ctx, int K, float *logp);
    int (*mult_put_rows)(void *ctx, const double *rows, int K);
    /* OPTIONAL pair (with the group above): the Multinomial log-marginals of a pass and of the merge candidates' pooled statistics on the
     * device -- dpmm_mult_master_pairs_ahead / dpmm_mult_master_marginals in dpmm_hip_master.h.  The engine asks ahead of step_stats for the pairs
     * whose gates are open and reads the results behind it; pairs it did not ask for, and steps after a split, are computed on the host. */
    int (*mult_pairs_ahead)(void *ctx, int outlier_first, const int32_t *ki, const int32_t *kj, int n);
    int (*mult_marginals)(void *ctx, int K, const double **rows_nl, const double **pairs_l, int *npairs);
    /* OPTIONAL pair (with the two above; may be NULL): dpmm_mult_master_rows_on_demand / dpmm_mult_master_rows_wait -- the engine declares once that
     * it reads step_stats' rows only through rows_wait (an accepted split / merge / removal, state access); the worker may then deliver them late. */
    int (*mult_rows_on_demand)(void *ctx, int on);
    int (*mult_rows_wait)(void *ctx);
} dpmmh_worker;

/* Options (dpmmh_model_set_option). */
enum {
    DPMMH_OPT_HARD_CLUSTERING = 1,   /* global_params.jl:8  -- argmax label assignment in every sweep */
    DPMMH_OPT_F32_QUIRK = 2,         /* utils.jl:66-72: accumulate log_multivariate_gamma in Float32 like the reference */
    DPMMH_OPT_THREADS = 3,           /* host threads for the per-distribution maths */
    DPMMH_OPT_SHARE_WORK = 4,        /* reserved (owner-computes sharing of the master's work across ranks): 0 = every rank computes everything
                                        (the only scheme built); any other value is refused */
    DPMMH_OPT_SPIN_US = 5,           /* bounded polling of the pool's workers between back-to-back parallel regions (default 150) */
    DPMMH_OPT_PREWAKE = 6,           /* 1 (default): wake the pool ~60 us before the statistics of a step are expected back */
    DPMMH_OPT_DEVICE_MASTER = 8,     /* NIW posteriors / factorisations / draws (Multinomial: the Dirichlet draws) on the worker's device: 1 on, 0 off, -1 (default): NIW for D >= 64, Multinomial for D >= 128 */
    DPMMH_OPT_DRAW_AHEAD = 9,        /* device master: launch the next parameter draws together with the posteriors (default 1); 0 draws when asked. Same results */
    DPMMH_OPT_NUMA_NODE = 7          /* >= 0: keep the pool's and the helper's threads on the CPUs of this NUMA node (the GPU's: worker numa_node); -1: leave them alone (default) */
};

int dpmmh_abi_version(void);

/* model_hyper_params (src/ds.jl:6-10) + the schedule constants the sweep needs.  burnout = burnout_period. */
int dpmmh_model_create(dpmmh_model **out, int prior_kind, int D, double alpha, int64_t n_total, uint64_t seed, int burnout,
                       int nthreads);
void dpmmh_model_destroy(dpmmh_model *m);
const char *dpmmh_model_last_error(const dpmmh_model *m);

/* distribution_hyper_params: niw_hyperparams(kappa, m, nu, psi) (priors/niw.jl:6-11) / multinomial_hyper(alpha)
 * (priors/multinomial_prior.jl:6-8).  which = 0: the cluster prior; 1: outlier_hyper_params (global_params.jl). */
int dpmmh_model_set_prior_niw(dpmmh_model *m, int which, double kappa, const double *mean, double nu, const double *psi);
int dpmmh_model_set_prior_mult(dpmmh_model *m, int which, const float *alpha);
/* outlier_mod > 0: cluster 1 is the outlier component (own prior, constant weight, never split / merged / removed). */
int dpmmh_model_set_outlier(dpmmh_model *m, double outlier_weight);
int dpmmh_model_set_option(dpmmh_model *m, int option, double value);
int dpmmh_model_bind_worker(dpmmh_model *m, const dpmmh_worker *w);
/* Called after the relabelling of accepted splits (and for every cluster at initialisation) when set: the smart-split
 * initialisation of sub-labels (smart_cluster_init!, local_clusters_actions.jl:555-627) lives in the host language. */
typedef int (*dpmmh_split_hook)(void *user, const int64_t *clusters_1based, int n);
int dpmmh_model_set_split_hook(dpmmh_model *m, dpmmh_split_hook hook, void *user);

/* init_model_from_data's labels (dp-parallel-sampling.jl:49-50) + init_first_clusters! (:62-78). */
int dpmmh_model_init_first_clusters(dpmmh_model *m, int init_clusters);
/* Resume / benchmark entry: the worker already holds labels for K clusters. */
int dpmmh_model_start_from_labels(dpmmh_model *m, int K);

/* group_step(group, no_more_splits, final) -- one restricted-Gibbs sweep. */
int dpmmh_group_step(dpmmh_model *m, int no_more_splits, int final);
/* The pieces of group_step, for hosts that interleave their own work (same order as the reference). */
int dpmmh_sample_clusters(dpmmh_model *m);
int dpmmh_update_suff_stats_posterior(dpmmh_model *m, const int64_t *clusters_1based, int n);   /* NULL: all (+ bad-cluster reset) */

/* calculate_posterior (dp-parallel-sampling.jl:458-470). */
double dpmmh_log_posterior(dpmmh_model *m);

/* State access, cluster order.  dpmmh_model_get copies a named field into `out` (capacity in bytes) and returns the
 * number of bytes of the field (negative on error; call with out = NULL to query the size).  Fields:
 *   "K" i64[1]            "counters" i64[8] (device epoch, draw epoch, split epoch, merge epoch, bad-cluster resets so far, steps with a reset, 0, 0)
 *   "N" f64[3K]           "sums" f64[3K][D]         "S" f64[3K][D][D]      (statistics; cluster row = left + right)
 *   "packed" f64[2K][stride]  (rows l, r of every cluster: the checkpoint form of the statistics)
 *   "kappa","nu","logdet_psi","log_marginal" f64[3K]   "m" f64[3K][D]   "U" f64[3K][D][D] (nu psi = U U')      -- NIW
 *   "alpha_post" f32[3K][D]                                                                                  -- Multinomial
 *   "mu" f32[3K][D]   "R" f32[3K][D][D]   "logdet" f32[3K]    /    "logp" f32[3K][D]     (last drawn parameters)
 *   "lr_weights" f32[K][2]   "weights" f32[K]   "splittable" u8[K]   "hist" f32[K][burnout+5]   "points_count" i64[K]
 *   "timers" f64[16] seconds accumulated per phase (see dpmmh_timer_names) */
int64_t dpmmh_model_get(dpmmh_model *m, const char *field, void *out, int64_t capacity);
/* Restore (checkpoints): set K first, then "packed", "lr_weights", "weights", "splittable", "hist", "points_count",
 * "counters", and optionally the parameter fields; posteriors are recomputed from the statistics. */
int dpmmh_model_set(dpmmh_model *m, const char *field, const void *in, int64_t bytes);
const char *dpmmh_timer_names(void);   /* comma-separated, in the order of "timers" */

/* Test hooks: the Metropolis ratios of the current state, in Float64.
 *   split: out[k] = log H of splitting cluster k (local_clusters_actions.jl:336-339), NaN where not eligible;
 *   merge: out[i*K + j] (i < j) = log H of merging (i, j) (shared_actions.jl:28-30), NaN where not eligible. */
int dpmmh_debug_split_log_hr(dpmmh_model *m, double *out);
int dpmmh_debug_merge_log_hr(dpmmh_model *m, double *out);

#ifdef __cplusplus
}
#endif
#endif /* DPMM_HOST_H */
