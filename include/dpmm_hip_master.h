/*
 * dpmm_hip_master.h -- optional companion of dpmm_hip.h: the MASTER's dense per-distribution maths on the worker's GPU.
 *
 * north_star keeps the posterior parameter draws and the split / merge Metropolis steps on the host, and the drop-in surface of
 * dpmm_hip.h is complete without this file (dpmm_step_stats -> host posteriors and draws -> dpmm_params_staging / dpmm_commit_params;
 * `DPMMH_OPT_DEVICE_MASTER = 0` of the engine, the `host_master` leg of bench.py).  These entry points are the opt-in fast path of
 * libdpmmhost.so's engine: the O(K D^3) posterior / factorisation / draw work (src/priors/niw.jl:20-40, multinomial_prior.jl:16-39) runs
 * next to the statistics it consumes, every DECISION (log-Hastings ratios, gates, accept / reject: src/local_clusters_actions.jl:318-413,
 * src/shared_actions.jl:12-66) stays with the caller.  A master that does not use them loses nothing but speed at D >= 64.
 */
#ifndef DPMM_HIP_MASTER_H
#define DPMM_HIP_MASTER_H

#include "dpmm_hip.h"

#define DPMM_MASTER_NSCALARS 8   /* doubles per distribution in the scalar records (dpmm_niw_master_posterior / _pairs, dpmm_step_master_device) */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the master's dense maths on the device (NIW prior; optional fast path for a master that otherwise works on the host) ----
 * The 3K posteriors, their factorisations and the parameter draws of a sweep are O(K D^3) and need the full packed rows: at
 * D = 256 that is 2-3 ms of host time and 30 MB over the host link per sweep.  With these calls the rows stay in HBM:
 *   dpmm_niw_master_setup       prior (kappa, nu, m [D], psi [D][D] row-major) -> device; enables the calls below
 *   dpmm_step_stats_device      dpmm_step_stats without the copy of the rows (*bad: [K] flags, pinned)
 *   dpmm_step_master_device     dpmm_step_stats_device + dpmm_niw_master_posterior for all K clusters (slots [K]) in one stream-ordered
 *                               sequence with ONE host wait.  draw_epoch != 0: the draws of dpmm_niw_master_draw(draw_epoch, K, slots, ...)
 *                               are launched as well, on a second stream, and the call returns when the POSTERIORS are done: the draws
 *                               need neither the weights nor the master's decisions and run while the host works.  The draw call uses
 *                               them if its epoch and slot map are the ones given here and no posterior changed in between (else it
 *                               draws again; results are the same either way -- the streams are keyed by epoch and position)
 *   dpmm_suffstats_device       dpmm_suffstats_host without the copy (rows of the listed clusters, 1-based; NULL = all)
 *   dpmm_niw_master_posterior   calc_posterior (src/priors/niw.jl:20-31) + factorisation nu' psi' = L' L for the listed clusters
 *                               (1-based) of the LAST statistics pass, stored under their slots (rows 3 slot + {0: cluster, 1: left,
 *                               2: right}); *small: pinned [n][3][DPMM_MASTER_NSCALARS] = {N, kappa', nu', log det(nu' psi') (NaN: not positive
 *                               definite), log Gamma_D(nu' / 2) (utils.jl:66-72: the D lgamma evaluations of a log-marginal), 3 spare}
 *   dpmm_niw_master_draw        sample_distribution (niw.jl:33-40) for all 3K distributions + the hand-over to the sweep kernels
 *                               (replaces dpmm_params_staging / dpmm_commit_params for this sweep): Sigma^-1 = R'R ~ Wishart(nu',
 *                               (nu' psi')^-1), mu ~ N(m', Sigma / kappa'); lr [K][2], w [K] as in dpmm_params_staging.  The random
 *                               streams are the library's own (Philox, keyed by seed, position in cluster order, epoch).
 *   dpmm_niw_master_pairs       pooled statistics of n slot pairs (check_and_merge!'s proposals, shared_actions.jl:21-27) -> *small: pinned
 *                               [n][DPMM_MASTER_NSCALARS], same record, of the pooled posterior under the cluster prior
 *   dpmm_niw_master_pairs_ahead the pairs the master MAY ask for after the next dpmm_step_master_device (all pairs of clusters whose merge
 *                               gate is open): that call computes them with the posteriors (same launch at D <= 128, second stream above) and
 *                               dpmm_niw_master_pairs answers from them (subset, any order) unless a slot got new statistics in between
 *   dpmm_niw_master_put_rows    rows [2K][1 + D + D(D+1)/2] from the host take the place of a statistics pass (restored state)
 *   dpmm_niw_master_rows        the stored statistics rows of the given slots -> out [n][2][1 + D + D(D+1)/2] (host)
 *   dpmm_niw_master_draws       the current draws in cluster order: mu [3K][D], R [3K][D][D] (upper triangular, full), logdet [3K] */
int dpmm_niw_master_setup(dpmm_ctx *ctx, double kappa, double nu, const double *m, const double *psi);
int dpmm_step_stats_device(dpmm_ctx *ctx, uint32_t reset_epoch, const uint8_t **bad);
int dpmm_step_master_device(dpmm_ctx *ctx, uint32_t reset_epoch, const int32_t *slots, uint32_t draw_epoch, const uint8_t **bad, const double **small);
int dpmm_suffstats_device(dpmm_ctx *ctx, const int64_t *cluster_idx, int n_idx);
int dpmm_niw_master_posterior(dpmm_ctx *ctx, const int64_t *clusters, const int32_t *slots, int n, const double **small);
int dpmm_niw_master_draw(dpmm_ctx *ctx, uint32_t epoch, int K, const int32_t *slot_of_cluster, const float *lr, const float *w);
int dpmm_niw_master_pairs_ahead(dpmm_ctx *ctx, const int32_t *slots_i, const int32_t *slots_j, int n);
int dpmm_niw_master_pairs(dpmm_ctx *ctx, const int32_t *slots_i, const int32_t *slots_j, int n, const double **small);
int dpmm_niw_master_put_rows(dpmm_ctx *ctx, const double *rows, int K);
int dpmm_niw_master_rows(dpmm_ctx *ctx, const int32_t *slots, int n, double *out);
int dpmm_niw_master_draws(dpmm_ctx *ctx, int K, float *mu, float *R, float *logdet);

/* ---- the Multinomial master's parameter draws on the device (optional, like the NIW group above) ----
 *   dpmm_mult_master_setup     prior alpha [D] (and the outlier component's prior, or NULL) -> device; enables the calls below
 *   dpmm_mult_master_draw      calc_posterior + sample_distribution (src/priors/multinomial_prior.jl:16-25) for all 3K distributions from the
 *                              rows the LAST full / per-step statistics pass left on the device (K must be that pass's K): alpha' = alpha +
 *                              Float32(sum x), log p = log Dirichlet(alpha') by Gamma variates (Philox keyed by seed, position in cluster
 *                              order, epoch), then the hand-over to the sweep kernels (replaces dpmm_params_staging / dpmm_commit_params for
 *                              this sweep); outlier_first: cluster 1 uses the outlier prior; lr [K][2], w [K] as in dpmm_params_staging
 *   dpmm_mult_master_draws     the current draws, log-probabilities [3K][D] (host)
 *   dpmm_mult_master_put_rows  rows [2K][1 + D] from the host take the place of a statistics pass (restored state) */
int dpmm_mult_master_setup(dpmm_ctx *ctx, const float *alpha, const float *alpha_outlier);
int dpmm_mult_master_draw(dpmm_ctx *ctx, uint32_t epoch, int K, int outlier_first, const float *lr, const float *w);
int dpmm_mult_master_draws(dpmm_ctx *ctx, int K, float *logp);
int dpmm_mult_master_put_rows(dpmm_ctx *ctx, const double *rows, int K);
/* The Multinomial master's log-marginals on the device (multinomial_prior.jl:34-39; check_and_merge!'s pooled statistics, LCA:385-413).
 * dpmm_mult_master_pairs_ahead: cluster pairs (0-based indices of the NEXT per-step pass) whose pooled log-marginal the master may ask for
 *   after that pass; the next dpmm_step_stats launches ONE kernel behind its statistics -- the 3K distributions of the pass + these pairs --
 *   and waits for it together with the rows.  n = 0 asks for the distributions only; more than 8192 pairs: none are computed.
 * dpmm_mult_master_marginals: rows_nl -> [3K][2] {N, log-marginal} (cluster, left, right per cluster), pairs_l -> [npairs] in the order they were
 *   asked for; pointers into a pinned block, valid until the next call that runs a statistics pass.  Without a pass-attached result for
 *   this K it computes now from the rows of the last full pass (dpmm_mult_master_put_rows counts as one) and waits; DPMM_ESTATE when those
 *   rows are not there (subset pass, split / merge / removal since). */
int dpmm_mult_master_pairs_ahead(dpmm_ctx *ctx, int outlier_first, const int32_t *ki, const int32_t *kj, int n);
int dpmm_mult_master_marginals(dpmm_ctx *ctx, int K, const double **rows_nl, const double **pairs_l, int *npairs);
/* A master that reads the rows of dpmm_step_stats only now and then (the log-marginals above decide a quiet step) says so once:
 * dpmm_mult_master_rows_on_demand(ctx, 1).  From then on, in a step whose draws are launched ahead (DPMM_OPT_MULT_DRAWS_AHEAD), dpmm_step_stats
 * returns when the FLAGS and the log-marginals are on the host; the rows follow behind the draws, and *packed must not be read before
 * dpmm_mult_master_rows_wait(ctx) has returned (a no-op in every other step).  Off by default: dpmm_step_stats then returns with the rows. */
int dpmm_mult_master_rows_on_demand(dpmm_ctx *ctx, int on);
int dpmm_mult_master_rows_wait(dpmm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
