/*
 * dpmm_hip_debug.h -- optional companion of dpmm_hip.h: diagnostics, timers and test hooks of libdpmmhip.so.
 * Nothing here is on the sweep's path; tests/ and bench.py are the callers.
 */
#ifndef DPMM_HIP_DEBUG_H
#define DPMM_HIP_DEBUG_H

#include "dpmm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics / parity: run the label phase of the last-set parameters and return the
 * Float32 table parr[k][i] = loglik_k(x_i) + log w_k WITHOUT the reference's constant
 * -D*D/2*log(2 pi) normaliser term (mv_gaussian.jl:24; identical for every cluster, so it
 * never affects a draw; add it back to compare with reference values).  out: [K][n_local]. */
int dpmm_debug_loglik(dpmm_ctx *ctx, float *out);
/* Same for the sub-label phase: out [2K][n_local], row 2k+s = loglik of every point under sub-cluster s of cluster k +
 * log lr_weights[k][s] -- for a point labelled k rows 2k, 2k+1 are exactly the two values create_subclusters_labels!
 * (local_clusters_actions.jl:83-95) draws from (same arithmetic as dpmm_sweep's sub-label phase). */
int dpmm_debug_subloglik(dpmm_ctx *ctx, float *out);

/* Diagnostic: the random inputs dpmm_niw_master_draw consumes for `epoch` and this cluster -> slot map, in cluster order: the Bartlett
 * factors A [3K][D][D] (lower triangular: chi_{nu' - r} on the diagonal, r = 0 .. D-1, standard normals below; Distributions.jl's
 * Wishart sampler behind niw.jl:35) and the mean normals xi [3K][D].  The draw is the deterministic function
 * R' = L^-1 A (nu' psi' = L'L, L lower), mu = m' + R^-1 xi / sqrt(kappa') of them: tests recompute it in Float64. */
int dpmm_debug_niw_draw_inputs(dpmm_ctx *ctx, uint32_t epoch, int K, const int32_t *slot_of_cluster, double *A, double *xi);

/* Reference bracket of the D <= 64 NIW sweep (DPMM_OPT_REF_BRACKET; csrc/niw_sweep.hip ref_bracket): for every point of the shard,
 * q_hi[i] = the bracket's certified upper end of q(x_i) = |R (x_i - mu)|^2 for the cluster-level factor of `cluster` (1-based), computed by
 * the two bf16 matrix passes the sweep uses, and q[i] = the Float32 evaluation it stands in for (sample_labels_worker!'s
 * log_likelihood!, src/local_clusters_actions.jl:112-134, src/distributions/mv_gaussian.jl:21-25).  The sweep's exactness argument is
 * q_hi[i] >= q[i] for every point; c_override > 0 replaces the library's rounding constant (tests show a too-small one failing).
 * NIW with D in 33..64, D % 4 == 0 and K > 2 only (DPMM_ESTATE otherwise). */
int dpmm_debug_ref_bracket(dpmm_ctx *ctx, int64_t cluster, float c_override, float *q_hi, float *q);
/* The pair-ball table of the parameters on the device (DPMM_OPT_PAIR_BALL; NIW, D in 33..64, 2 <= K <= 256): pd [K][K], pd[k K + j] = the
 * tabulated lower bound of |R_j (mu_k - mu_j)|; sn [K] = the tabulated upper bounds of |R_j|_2.  Tests check both against Float64 values. */
int dpmm_debug_pair_ball(dpmm_ctx *ctx, float *pd, float *sn);
/* D = 65 .. 256 (the LDS-staged sweep kernels): the reference bracket runs as a launch of its own in front of the sweep (niw_bracket_big_kernel).
   This runs it on the current labels and parameters and returns what the sweep would read: tile_flags[ceil(n / 128)] = 1 + k0 (0-based k0)
   for a 128-point tile of the visiting order whose points all carry label k0 + 1, else 0; aref[n] = per POSITION of the visiting order
   (storage order before the first statistics pass) the bracket's lower end of a_k0 = cst - q_hi / 2 (tiles with flag 0: 0). */
int dpmm_debug_bracket_big(dpmm_ctx *ctx, float *aref, uint32_t *tile_flags);
/* Multinomial device master: how many dpmm_mult_master_draw calls took the draws dpmm_step_stats had launched ahead (DPMM_OPT_MULT_DRAWS_AHEAD). */
int dpmm_debug_mult_draws_ahead(dpmm_ctx *ctx, long long *used);

/* Milliseconds spent in the dominant kernels during the last dpmm_sweep /
 * dpmm_suffstats_* call, measured with HIP events on the ctx stream (0 if none yet, or when DPMM_OPT_KERNEL_TIMING is off -- the default).
 * Calling this waits for the measured kernels (the closing event of each pair), not for later work on the stream. */
int dpmm_last_kernel_ms(dpmm_ctx *ctx, float *sweep_ms, float *suffstats_ms);
/* D in 33..64 with the bf16 sub-label evaluation active (DPMM_OPT_B3_SUBLABELS): a sweep is niw_lean_kernel (tiles the cheap screens settle) + the
 * sweep kernel on the spans it hands on (labels and sub-labels in one launch), or -- without the lean launch -- the sweep kernel in its labels-only
 * form + niw_sub_kernel on every tile.  With DPMM_OPT_KERNEL_TIMING bits 0 and 3 set: out3 = milliseconds of {lean launch, sweep-kernel launch,
 * what follows it up to the end of the sweep (niw_sub_kernel, if it ran)} of the last dpmm_sweep (0 for a part that did not run; all 0 for any other
 * kind of sweep). */
int dpmm_last_sweep_parts_ms(dpmm_ctx *ctx, float *out3);
/* Work the dpmm_sweep calls (NIW) since the previous call really executed, counted on the device (a slot per wave, no atomics):
 *   TOTALS over out16[7] launches of: out16[0] wave tiles, [1] full quadratic-form evaluations (per wave), [2] Float32 16-row MFMA screens
 *   (per wave; an evaluation left after its first row block counts as four), [3] tail-screened cluster pairs (per wave), [8] reference
 *   brackets, [11] bf16 bottom screens, [13] bf16 top screens (per wave: bf16 matrix work, NOT part of the Float32 figure);
 *   [4] Float32 matrix instructions per full evaluation, [5] per 16-row screen, [6] flops per Float32 matrix instruction
 *   (v_mfma_f32_16x16x4_f32: 2048), [9] / [12] / [14] bf16 matrix instructions per bracket / bottom screen / top screen, [10] flops per
 *   bf16 matrix instruction (v_mfma_f32_16x16x32_bf16: 16384), [15] low 32 bits: direction screens, high 32 bits: bf16 three-plane
 *   evaluations of sub-cluster matrices (DPMM_OPT_B3_SUBLABELS: 144 bf16 matrix instructions + 4 Float32 row sums each).  Executed Float32
 *   matrix flops of those launches = (out16[1]*out16[4] + out16[2]*out16[5] + 4 * (out16[15] >> 32)) * out16[6] -- the figure
 *   SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 reports (summed over the kernels of a sweep).  The counters are cleared;
 *   calling this synchronises the stream (a benchmark calls it once after its timed loop, not once per step). */
int dpmm_last_sweep_work(dpmm_ctx *ctx, uint64_t *out16);
/* HIP-event time of the last all-reduce of each kind on the ctx stream (0 if none; synchronises the stream); DPMM_OPT_KERNEL_TIMING bit 4. */
int dpmm_last_comm_ms(dpmm_ctx *ctx, float *counts_ms, float *rows_ms);

/* Health counters of this ctx since creation (n >= 1 entries written, the rest 0):
 *   out[0] = dpmm_step_master_device calls whose event wait returned before the posteriors' records had reached host memory (the call
 *            then waits them out; a non-zero count is a runtime / driver anomaly worth reporting, the results are unaffected). */
int dpmm_debug_counters(dpmm_ctx *ctx, int64_t *out, int n);
/* Test hook, process-wide: fn(arg) is called on the host in front of every kernel launch of the library (fn == NULL: off, the default).
 * tests/tools/poison.py uses it to refill LDS and the register files with a NaN pattern between the library's own kernels
 * (tests/test_gpu_uninit.py); fn may synchronise the device and launch kernels of its own on other streams. */
int dpmm_debug_set_prelaunch_hook(void (*fn)(void *), void *arg);

#ifdef __cplusplus
}
#endif
#endif
