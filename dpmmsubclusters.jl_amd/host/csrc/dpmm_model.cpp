// dpmm_model.cpp -- the master half of one restricted-Gibbs sweep, native (libdpmmhost.so; contract: include/dpmm_host.h).
//
// Restates, over slot-indexed struct-of-arrays cluster state and a table of worker entry points, the master-process
// functions of the reference (paths relative to the reference checkout):
//
//     group_step                          src/local_clusters_actions.jl:658-673
//     sample_clusters!                    src/local_clusters_actions.jl:417-437
//     sample_cluster_params               src/shared_actions.jl:41-66   (burn-in gate :51-63)
//     update_suff_stats_posterior!        src/local_clusters_actions.jl:206-254
//     reset_bad_clusters!                 src/local_clusters_actions.jl:501-516
//     check_and_split! / should_split_local! / split_cluster_local!   :345-382, :318-343, :280-291
//     check_and_merge! / should_merge! / merge_clusters!              :385-413, shared_actions.jl:21-38, :308-315
//     remove_empty_clusters!              src/local_clusters_actions.jl:457-471
//     init_first_clusters!                src/dp-parallel-sampling.jl:62-78
//     calculate_posterior                 src/dp-parallel-sampling.jl:458-470
//
// Layout.  A cluster owns a SLOT for life; `slot[k]` maps the k-th live cluster to it, so splits, merges and compaction
// never move the big per-distribution arrays (a posterior factor is D*D doubles; the worker's parameter staging D*D floats).
// Row 3*slot + w is distribution w (0 cluster, 1 left, 2 right).  Statistics are kept in the PACKED form the worker
// delivers ({N, sum, lower triangle of S} for the left and the right sub-cluster; cluster = left + right) and are never
// expanded to full matrices: posteriors, factorisations and draws read them directly.
#include "../../../include/dpmm_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <sys/prctl.h>

#include <algorithm>
#include <string>
#include <vector>

#include "hostlib.h"
#include "hostmath.h"

using dpmmh::Philox;
using dpmmh::Pool;

namespace {

enum Timer { T_SAMPLE = 0, T_MISC, T_COMMIT, T_SWEEP_LAUNCH, T_STATS_WAIT, T_POSTERIOR, T_SPLIT, T_MERGE, T_MERGE_PAIRS, T_REMOVE, T_HOOK, T_NOISE_WAIT, T_EXCHANGE, T_COUNT };
const char *kTimerNames = "sample_params,host_misc,commit_params,sweep_launch,stats_wait,posterior,split,merge,merge_pairs,remove_empty,split_hook,noise_wait,exchange";

inline double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

struct NiwPrior {
    bool set = false;
    double kappa = 0, nu = 0, logdet_psi = 0;
    std::vector<double> m, psi, psi_lo;   // psi_lo: symmetrised, packed lower triangle (hostmath.h)
    double lmg0[2] = {0, 0};              // log_multivariate_gamma(nu / 2) in Float64 / Float32-quirk arithmetic
};
struct MultPrior {
    bool set = false;
    std::vector<float> alpha;
    double alpha_sum = 0.0, lg_alpha_sum = 0.0;      // sum alpha_d, sum lgamma(alpha_d): constants of every log-marginal under this prior
};

// One persistent helper thread: runs a job (the noise generation) while the calling thread blocks on the GPU.
class Helper {
  public:
    ~Helper() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    void submit(std::function<void()> job) {
        wait();
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!th_.joinable()) {
                th_ = std::thread([this] { loop(); });
                if (have_affinity_) pthread_setaffinity_np(th_.native_handle(), sizeof(affinity_), &affinity_);
            }
            job_ = std::move(job); busy_ = true;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return !busy_; });
    }
    // for jobs: sleep until the CLOCK_MONOTONIC time `t_abs` (seconds) unless cancel() comes first; true = the time was reached
    bool sleep_until(double t_abs) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            if (cancel_ || stop_) return false;
            const double left = t_abs - now_s();
            if (left <= 0) return true;
            cv_.wait_for(lk, std::chrono::duration<double>(left));
        }
    }
    void cancel() {
        { std::lock_guard<std::mutex> lk(mu_); cancel_ = true; }
        cv_.notify_all();
    }
    void set_numa_node(int node) {
        cpu_set_t set;
        if (!Pool::node_cpus(node, &set)) return;
        std::lock_guard<std::mutex> lk(mu_);
        affinity_ = set; have_affinity_ = true;
        if (th_.joinable()) pthread_setaffinity_np(th_.native_handle(), sizeof(set), &set);
    }
  private:
    void loop() {
        prctl(PR_SET_TIMERSLACK, 1000UL);     // 1 us instead of the default 50 us: sleep_until is used for sub-100-us scheduling
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (busy_ && job_); });
                if (stop_) return;
                job = std::move(job_); job_ = nullptr; cancel_ = false;
            }
            job();
            { std::lock_guard<std::mutex> lk(mu_); busy_ = false; }
            done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::thread th_;
    std::function<void()> job_;
    bool busy_ = false, stop_ = false, cancel_ = false, have_affinity_ = false;
    cpu_set_t affinity_;
};

// random streams of the master side (the worker uses 0..3, the draws 16..18)
enum : uint32_t { ST_LR = 20, ST_WEIGHTS = 21, ST_SPLIT = 22, ST_MERGE = 23 };

}  // namespace

struct dpmmh_model {
    int kind = 0, D = 0, burnout = 0, hist_len = 0, nthreads = 1;
    int64_t T = 0, stride = 0, n_total = 0;
    double alpha = 0;
    uint64_t seed = 0;
    NiwPrior niw[2];
    MultPrior mult[2];
    double outlier_weight = 0;
    bool hard = false, f32_quirk = false;
    int share_work = 0;
    dpmmh_worker W{};
    bool bound = false;
    dpmmh_split_hook hook = nullptr;
    void *hook_user = nullptr;

    int K = 0, cap = 0;
    std::vector<int> slot;            // [K] cluster -> slot
    std::vector<uint8_t> slot_used;   // [cap]
    // per slot
    std::vector<double> packed;       // [cap][2][stride]
    std::vector<double> kappa, nu, ldpsi, L, Nrow;   // [3 cap]
    std::vector<double> mean, U;      // [3 cap][D], [3 cap][D*D]   (NIW; U holds L = U' of nu psi = U U' = L' L, row-major lower)
    std::vector<float> apost;         // [3 cap][D]                 (Multinomial)
    std::vector<uint8_t> splittable;  // [cap]
    std::vector<float> hist;          // [cap][hist_len]
    std::vector<int64_t> points_count;   // [cap]
    // worker staging (slot-indexed mu / mat / logdet; cluster-ordered lr / w / slot map)
    float *st_mu = nullptr, *st_mat = nullptr, *st_logdet = nullptr, *st_lr = nullptr, *st_w = nullptr;
    int32_t *st_slot = nullptr;
    int st_slots = 0;

    uint32_t epoch = 0;               // device-side randomised calls
    uint32_t draw_epoch = 1u << 20;   // parameter draws (predictable: the noise is generated ahead)
    uint32_t split_epoch = 0, merge_epoch = 0;
    int64_t bad_total = 0, bad_steps = 0;   // diagnostics: clusters reset as "bad" so far, steps with any

    // noise generated while the GPU sweeps
    Helper helper;
    // device master (worker.niw_*): dev_state = the device holds the posteriors of every live slot; host_dense = the host's packed
    // rows / means / factors are current (false while the device is the only one that has seen the latest statistics)
    static constexpr int kDevScalars = 8;      // DPMM_MASTER_NSCALARS of the worker ABI: N, kappa', nu', log det(nu' psi'), log Gamma_D(nu' / 2), spare
    int opt_dev_master = -1;
    int opt_draw_ahead = 1;          // device master: the next draws are launched with the posteriors (DPMMH_OPT_DRAW_AHEAD)
    bool dev_pairs_ok = false;
    bool dev_setup = false, dev_state = false, host_dense = true, host_rows = true, dev_draw = false;   // host_rows: the packed rows alone are current
    // Multinomial (worker.mult_*): the Dirichlet draws happen on the device while the worker's last statistics pass holds the rows of
    // all K clusters as the engine knows them (mult_rows_current: set by a full pass, cleared by an accepted split / merge / removal)
    bool mult_dev_setup = false, mult_dev_failed = false, mult_rows_current = false;
    // ... and its log-marginals (worker.mult_pairs_ahead / mult_marginals): the pairs asked for ahead of the step's statistics pass (cluster
    // indices), their pooled log-marginals as the worker returned them, valid until a split / merge / removal changes the clusters
    std::vector<int32_t> mp_i, mp_j;
    std::vector<double> mp_L;
    bool mp_valid = false;
    bool prewake = true;
    double wait_ema = 0.0, t_stats_back = 0.0;
    static constexpr double kPrewakeLead = 60e-6;   // seconds before the predicted hand-back
    bool noise_pending = false;
    uint32_t noise_epoch = 0;
    int noise_rows = 0;
    std::vector<double> noise_A, noise_xi, noise_cx, noise_cu;   // NIW: Bartlett normals, mean normals, chi first trials; Multinomial: first-trial normals / uniforms

    double timers[16] = {0};
    std::string err;

    // ---------------------------------------------------------------- helpers
    int fail(const std::string &msg) { err = msg; return -1; }
    int wfail(const char *what) {
        const char *e = (W.last_error && W.ctx) ? W.last_error(W.ctx) : nullptr;
        err = std::string(what) + " failed" + (e && *e ? std::string(": ") + e : std::string());
        return -2;
    }
    const NiwPrior &niw_of(int s) const { return (outlier_weight > 0 && K > 0 && s == slot[0] && niw[1].set) ? niw[1] : niw[0]; }
    const MultPrior &mult_of(int s) const { return (outlier_weight > 0 && K > 0 && s == slot[0] && mult[1].set) ? mult[1] : mult[0]; }
    double *prow(int s, int side) { return packed.data() + ((size_t)s * 2 + side) * stride; }
    double Nl(int s) { return Nrow[3 * s + 1]; }      // (from the posterior scalars: valid whether the rows live on the host or the device)
    double Nr(int s) { return Nrow[3 * s + 2]; }
    double Nc(int s) { return Nrow[3 * s + 1] + Nrow[3 * s + 2]; }
    uint32_t next_epoch() { return ++epoch; }
    bool has_outlier() const { return outlier_weight > 0; }

    int ensure_slots(int need) {
        if (need <= cap && st_slots >= cap) return 0;
        int ncap = std::max(8, cap);
        while (ncap < need) ncap *= 2;
        const size_t DD = (size_t)D * D;
        slot_used.resize(ncap, 0);
        packed.resize((size_t)ncap * 2 * stride, 0.0);
        for (auto *v : {&kappa, &nu, &ldpsi, &L, &Nrow}) v->resize((size_t)ncap * 3, 0.0);
        if (kind == DPMMH_PRIOR_NIW) { mean.resize((size_t)ncap * 3 * D, 0.0); U.resize((size_t)ncap * 3 * DD, 0.0); }
        else apost.resize((size_t)ncap * 3 * D, 0.f);
        splittable.resize(ncap, 0);
        hist.resize((size_t)ncap * hist_len, -INFINITY);
        points_count.resize(ncap, 0);
        cap = ncap;
        if (bound) {
            if (W.params_staging(W.ctx, cap, &st_mu, &st_mat, &st_logdet, &st_lr, &st_w, &st_slot)) return wfail("params_staging");
            st_slots = cap;
        }
        return 0;
    }
    int alloc_slot() {
        for (int s = 0; s < cap; ++s)
            if (!slot_used[s]) { reset_slot(s); slot_used[s] = 1; return s; }
        const int s = cap;
        if (ensure_slots(cap + 1)) return -1;
        reset_slot(s); slot_used[s] = 1;
        return s;
    }
    void reset_slot(int s) {
        std::fill(prow(s, 0), prow(s, 0) + 2 * stride, 0.0);
        splittable[s] = 0;
        std::fill(hist.begin() + (size_t)s * hist_len, hist.begin() + (size_t)(s + 1) * hist_len, -INFINITY);
        points_count[s] = 0;
    }
    void reset_hist(int s) { std::fill(hist.begin() + (size_t)s * hist_len, hist.begin() + (size_t)(s + 1) * hist_len, -INFINITY); }

    // ---------------------------------------------------------------- posterior of one row (calc_posterior + factorisation + marginal)
    // src rows: the statistic set is sum_i coef[i] * rows[i] (packed rows).  Scratch P: D*D doubles.
    void niw_row(int s, int w, const double *l, const double *r) {
        const NiwPrior &pr = niw_of(s);
        const int row = 3 * s + w;
        const double cl = (w != 2) ? 1.0 : 0.0, cr = (w != 1) ? 1.0 : 0.0;
        const size_t DD = (size_t)D * D;
        double *P = U.data() + (size_t)row * DD;               // the scale matrix is built and factorised where the factor lives
        const double N = dpmmh::niw_posterior_packed(D, pr.kappa, pr.nu, pr.m.data(), pr.psi_lo.data(), l, r, cl, cr, &kappa[row], &nu[row],
                                                     mean.data() + (size_t)row * D, P);
        Nrow[row] = N;
        const double ld = dpmmh::chol_ltl(P, D, nullptr);      // nu' psi' = L' L in place (L = U', row-major lower; entries above the diagonal
                                                               // are scratch, nobody reads them); NaN when not positive definite
        ldpsi[row] = ld - D * log(nu[row]);
        L[row] = niw_marginal(pr, kappa[row], nu[row], ldpsi[row], N);
    }
    double niw_marginal(const NiwPrior &pr, double k1, double v1, double ld1, double N) const {
        // log Gamma_D(nu0 / 2) is a constant of the prior (D lgamma evaluations): computed when the prior is set, for both arithmetic modes
        return dpmmh::niw_log_marginal(D, pr.kappa, pr.nu, pr.logdet_psi, pr.lmg0[f32_quirk ? 1 : 0], k1, v1, ld1, N, f32_quirk);
    }
    void mult_row(int s, int w, const double *l, const double *r) {
        const MultPrior &pr = mult_of(s);
        const int row = 3 * s + w;
        const double cl = (w != 2) ? 1.0 : 0.0, cr = (w != 1) ? 1.0 : 0.0;
        const double N = cl * l[0] + cr * r[0];
        Nrow[row] = N;
        float *ap = apost.data() + (size_t)row * D;
        if (N == 0.0) {
            memcpy(ap, pr.alpha.data(), sizeof(float) * D);
        } else {   // multinomial_prior.jl:16-21: alpha' = alpha + sum x, the sum held as Float32 by the reference
            for (int d = 0; d < D; ++d) ap[d] = pr.alpha[d] + (float)(cl * l[1 + d] + cr * r[1 + d]);
        }
        L[row] = dpmmh::mult_log_marginal(D, pr.alpha.data(), ap);
    }

    // ---------------------------------------------------------------- device master
    bool use_dev() {
        if (kind != DPMMH_PRIOR_NIW || !W.niw_master_setup || !W.step_stats_device || !W.step_master_device || !W.stats_device || !W.niw_posterior || !W.niw_draw ||
            !W.niw_pairs || !W.niw_put_rows || !W.niw_rows || !W.niw_draws || has_outlier() || D > 256)
            return false;
        if (!(opt_dev_master == 1 || (opt_dev_master < 0 && D >= 64))) return false;
        if (!dev_setup) {
            const NiwPrior &pr = niw[0];
            if (W.niw_master_setup(W.ctx, pr.kappa, pr.nu, pr.m.data(), pr.psi.data())) { wfail("niw_master_setup"); return false; }
            dev_setup = true;
        }
        return true;
    }
    bool use_mult_dev() {
        if (kind != DPMMH_PRIOR_MULT || mult_dev_failed || !W.mult_master_setup || !W.mult_draw || !W.mult_draws || !W.mult_put_rows) return false;
        if (!(opt_dev_master == 1 || (opt_dev_master < 0 && D >= 128))) return false;
        if (!mult_dev_setup) {
            if (W.mult_master_setup(W.ctx, mult[0].alpha.data(), mult[1].set ? mult[1].alpha.data() : nullptr)) { wfail("mult_master_setup"); mult_dev_failed = true; return false; }
            mult_dev_setup = true;
            // (this engine reads a step's rows only through pull_rows: the worker may deliver them behind the draws it launches ahead)
            if (W.mult_rows_on_demand && W.mult_rows_wait && W.mult_pairs_ahead && W.mult_marginals) W.mult_rows_on_demand(W.ctx, 1);
        }
        return true;
    }
    // posteriors + factorisations of clusters `ks` on the device (from the statistics pass that just ran); scalars come back
    int ingest_device(const std::vector<int> &ks) {
        const int n = (int)ks.size();
        std::vector<int64_t> cl(n);
        std::vector<int32_t> sl(n);
        for (int i = 0; i < n; ++i) { cl[i] = ks[i] + 1; sl[i] = slot[ks[i]]; }
        const double *sm = nullptr;
        if (W.niw_posterior(W.ctx, cl.data(), sl.data(), n, &sm)) return wfail("niw_posterior");
        return apply_device_posteriors(sl, sm);
    }
    // the scalars of n clusters' posteriors (slots sl, rows sm [n][3][8]) -> the engine's per-row state + log-marginals
    int apply_device_posteriors(const std::vector<int32_t> &sl, const double *sm) {
        const int n = (int)sl.size();
        const NiwPrior &pr = niw[0];
        // the worker returns log Gamma_D(nu' / 2) with the other scalars (the D lgamma evaluations of a log-marginal): what is left is a
        // handful of logarithms per distribution, done here -- no pool job, no wake-up (on a busy host that cost up to 0.15 ms per step).
        // The Float32-accumulator mode of the reference (DPMMH_OPT_F32_QUIRK) recomputes the term on the host.
        auto one = [&](int item) {
            const int i = item / 3, w = item % 3;
            const double *o = sm + (size_t)item * kDevScalars;
            const int row = 3 * sl[i] + w;
            Nrow[row] = o[0]; kappa[row] = o[1]; nu[row] = o[2];
            ldpsi[row] = o[3] - D * log(nu[row]);
            L[row] = f32_quirk ? niw_marginal(pr, kappa[row], nu[row], ldpsi[row], o[0])
                               : dpmmh::niw_log_marginal_lmg(D, pr.kappa, pr.nu, pr.logdet_psi, pr.lmg0[0], kappa[row], nu[row], ldpsi[row], o[0], o[4]);
        };
        if (f32_quirk) Pool::get().run(3 * n, nthreads, [&](int item, int) { one(item); });
        else for (int item = 0; item < 3 * n; ++item) one(item);
        for (int i = 0; i < n; ++i) points_count[sl[i]] = (int64_t)llrint(Nrow[3 * sl[i]]);
        host_dense = false; host_rows = false;
        return 0;
    }
    // bring the host's packed rows up to date from the device's statistics (merge proposals, state access) ...
    int pull_rows() {
        if (host_rows) return 0;
        if (kind == DPMMH_PRIOR_MULT) {
            if (!mult_pending) return fail("the Multinomial rows of the last statistics pass are gone");
            if (W.mult_rows_wait && W.mult_rows_wait(W.ctx)) return wfail("mult_rows_wait");
            for (int k = 0; k < K; ++k) memcpy(prow(slot[k], 0), mult_pending + (size_t)(2 * k) * stride, sizeof(double) * 2 * stride);
            mult_pending = nullptr;
            host_rows = true;
            return 0;
        }
        std::vector<int32_t> sl(K);
        for (int k = 0; k < K; ++k) sl[k] = slot[k];
        std::vector<double> rows((size_t)K * 2 * stride);
        if (W.niw_rows(W.ctx, sl.data(), K, rows.data())) return wfail("niw_rows");
        for (int k = 0; k < K; ++k) memcpy(prow(slot[k], 0), rows.data() + (size_t)k * 2 * stride, sizeof(double) * 2 * stride);
        host_rows = true;
        return 0;
    }
    // ... and the posteriors (means, factors) computed from them on the host (host draws, accepted merges, access to m / U)
    int pull_state() {
        if (host_dense) return 0;
        if (int rc = pull_rows()) return rc;
        std::vector<double> none;
        Pool::get().run(3 * K, nthreads, [&](int item, int) { refresh_row(slot[item / 3], item % 3, none); });
        host_dense = true;
        return 0;
    }

    // statistics of clusters `ks` (cluster order) arrive as packed rows src[2k], src[2k+1]: store + posteriors
    void ingest(const double *src, const std::vector<int> &ks) {
        const int n = (int)ks.size();
        Pool::get().run(3 * n, nthreads, [&](int item, int) {
            const int k = ks[item / 3], w = item % 3, s = slot[k];
            const double *l = src + (size_t)(2 * k) * stride, *r = l + stride;
            if (w == 1) memcpy(prow(s, 0), l, sizeof(double) * stride);
            if (w == 2) memcpy(prow(s, 1), r, sizeof(double) * stride);
            if (kind == DPMMH_PRIOR_NIW) {
                niw_row(s, w, l, r);
            } else {
                mult_row(s, w, l, r);
            }
        });
        for (int k : ks) {
            const int s = slot[k];
            points_count[s] = (int64_t)llrint(Nrow[3 * s]);
        }
    }
    // Multinomial, device marginals: rows to the slots, N and the log-marginals as the worker computed them; the Float32 posterior
    // parameters (apost) are formed only when somebody needs them (host draws, state access: pull_state)
    // The rows themselves stay in the worker's pinned block (`mult_pending`) until somebody asks (pull_rows: an accepted split or merge, a pair the
    // worker did not evaluate, state access): 2K rows of D + 1 doubles are half a megabyte at D = 1000, 22 us of every step spent copying
    // what a steady-state step never reads.  The block is the worker's until its next statistics call (update_all_with_reset drops the pointer,
    // update_subset pulls first).
    const double *mult_pending = nullptr;
    void ingest_mult_marginals(const double *src, const double *nl) {
        for (int k = 0; k < K; ++k) {
            const int s = slot[k];
            for (int w = 0; w < 3; ++w) { Nrow[3 * s + w] = nl[2 * (3 * k + w)]; L[3 * s + w] = nl[2 * (3 * k + w) + 1]; }
            points_count[s] = (int64_t)llrint(Nrow[3 * s]);
        }
        mult_pending = src;
        host_dense = false; host_rows = false;
    }
    // recompute the three posteriors of slot s from its stored statistics
    void refresh_slot(int s, std::vector<double> &sc) {
        for (int w = 0; w < 3; ++w) refresh_row(s, w, sc);
    }
    void refresh_row(int s, int w, std::vector<double> &) {
        if (kind == DPMMH_PRIOR_NIW) {
            niw_row(s, w, prow(s, 0), prow(s, 1));
        } else {
            mult_row(s, w, prow(s, 0), prow(s, 1));
        }
    }
    void copy_row_post(int dst, int src) {
        kappa[dst] = kappa[src]; nu[dst] = nu[src]; ldpsi[dst] = ldpsi[src]; L[dst] = L[src]; Nrow[dst] = Nrow[src];
        if (kind == DPMMH_PRIOR_NIW) {
            const size_t DD = (size_t)D * D;
            memcpy(mean.data() + (size_t)dst * D, mean.data() + (size_t)src * D, sizeof(double) * D);
            memcpy(U.data() + (size_t)dst * DD, U.data() + (size_t)src * DD, sizeof(double) * DD);
        } else {
            memcpy(apost.data() + (size_t)dst * D, apost.data() + (size_t)src * D, sizeof(float) * D);
        }
    }

    // ---------------------------------------------------------------- noise (runs while the GPU sweeps)
    // ... and wakes the pool shortly before the statistics are expected back (prediction: the previous steps' launch-to-statistics time)
    void start_noise(double t_launch = 0.0) {
        wait_noise();
        const bool niw_dev = dev_draw;       // device draws (either prior) make their own noise (a wrong guess costs an inline generation)
        const bool niw_noise = kind == DPMMH_PRIOR_NIW && !niw_dev;
        // (no pool job follows the statistics on the device-master path -- the worker returns the lgamma terms too -- so nobody is woken)
        const bool pool_after_stats = !(kind == DPMMH_PRIOR_NIW && niw_dev && !f32_quirk);
        const double pre_at = (prewake && pool_after_stats && nthreads > 1 && t_launch > 0.0 && wait_ema > 0.0) ? t_launch + wait_ema - kPrewakeLead : 0.0;
        const int rows = 3 * (K + 4);   // head-room for clusters born from splits
        const size_t DD = (size_t)D * D;
        if (niw_noise) {
            if (noise_A.size() < (size_t)rows * DD) {
                noise_A.resize((size_t)rows * DD); noise_xi.resize((size_t)rows * D);
                noise_cx.resize((size_t)rows * D); noise_cu.resize((size_t)rows * D);
            }
        } else {
            if (noise_A.size() < (size_t)rows * D) { noise_A.resize((size_t)rows * D); noise_xi.resize((size_t)rows * D); }   // first-trial normals / uniforms
        }
        noise_epoch = draw_epoch + 1; noise_rows = niw_dev ? 0 : rows;
        // nothing to generate and nobody to wake (the device-master path): no helper job at all -- the next step would otherwise wait for
        // the helper thread to have RUN its empty job, and one late wake-up of that thread was a 2.7 ms stall of a 2 ms step
        if (niw_dev && pre_at <= 0.0) return;
        const int nt = nthreads;
        helper.submit([this, rows, DD, nt, niw_noise, niw_dev, pre_at] {
            if (!niw_dev) Pool::get().run(rows, nt, [&](int i, int) {
                if (niw_noise) dpmmh::niw_noise_one(D, seed, (uint32_t)i, noise_epoch, noise_A.data() + (size_t)i * DD, noise_xi.data() + (size_t)i * D,
                                                    noise_cx.data() + (size_t)i * D, noise_cu.data() + (size_t)i * D);
                else dpmmh::dirichlet_noise_one(D, seed, (uint32_t)i, noise_epoch, noise_A.data() + (size_t)i * D, noise_xi.data() + (size_t)i * D);
            });
            if (pre_at > 0.0 && helper.sleep_until(pre_at)) Pool::get().prewake();
        });
        noise_pending = true;
    }
    void wait_noise() {
        if (noise_pending) { helper.wait(); noise_pending = false; }
    }

    // ---------------------------------------------------------------- step 1: sample_clusters! (LCA:417-437, SA:41-66)
    int sample_clusters() {
        if (!bound) return fail("no worker bound");
        if (K < 1) return fail("no clusters");
        double t0 = now_s();
        wait_noise();
        timers[T_NOISE_WAIT] += now_s() - t0; t0 = now_s();
        draw_epoch += 1;
        const bool have_noise = noise_epoch == draw_epoch && noise_rows > 0;
        const size_t DD = (size_t)D * D;
        const bool mdev = kind == DPMMH_PRIOR_MULT && mult_rows_current && use_mult_dev();
        const bool dev = mdev || (kind == DPMMH_PRIOR_NIW && use_dev() && dev_state);     // the draws happen on the device, after the weights below
        if (!dev) { if (int rc = pull_state()) return rc; }
        std::vector<std::vector<double>> scratch(std::max(1, nthreads));
        if (!dev) Pool::get().run(3 * K, nthreads, [&](int id, int th) {
            const int k = id / 3, w = id % 3, row = 3 * slot[k] + w;
            if (kind == DPMMH_PRIOR_NIW) {
                auto &sc = scratch[th];
                if (sc.empty()) sc.resize(dpmmh::niw_draw_scratch_doubles(D));
                const bool pre = have_noise && id < noise_rows;
                dpmmh::niw_draw_one(D, kappa[row], nu[row], mean.data() + (size_t)row * D, U.data() + (size_t)row * DD, seed, (uint32_t)id,
                                    draw_epoch, pre ? noise_A.data() + (size_t)id * DD : nullptr, pre ? noise_xi.data() + (size_t)id * D : nullptr,
                                    sc.data(), st_mu + (size_t)row * D, st_mat + (size_t)row * T, &st_logdet[row], true,
                                    pre ? noise_A.data() + (size_t)id * DD : nullptr,        // solved in place: the helper refills it next step
                                    pre ? noise_cx.data() + (size_t)id * D : nullptr, pre ? noise_cu.data() + (size_t)id * D : nullptr);
            } else {
                auto &sc = scratch[th];
                if (sc.empty()) sc.resize(D);
                const bool pre = have_noise && id < noise_rows;
                dpmmh::dirichlet_log_one(D, apost.data() + (size_t)row * D, seed, (uint32_t)id, draw_epoch, sc.data(), st_mat + (size_t)row * D,
                                         pre ? noise_A.data() + (size_t)id * D : nullptr, pre ? noise_xi.data() + (size_t)id * D : nullptr);
            }
        });
        timers[T_SAMPLE] += now_s() - t0; t0 = now_s();
        // lr_weights ~ Dirichlet(N_l + alpha/2, N_r + alpha/2); burn-in gate; mixture weights
        const int b = burnout;
        for (int k = 0; k < K; ++k) {
            const int s = slot[k];
            dpmmh::dirichlet2(Nl(s) + alpha / 2, Nr(s) + alpha / 2, seed, (uint32_t)k, draw_epoch, ST_LR, st_lr + 2 * k);
            float *h = hist.data() + (size_t)s * hist_len;
            for (int i = 0; i + 1 < b; ++i) h[i] = h[i + 1];
            h[b - 1] = (float)(L[3 * s + 1] + L[3 * s + 2]);
            double now = 0.0;
            for (int i = 0; i < b; ++i) now += (double)h[i] * (1.0 / (b - 0.1));
            if (now != -INFINITY && (now - (double)h[b - 1]) < 1e-2) splittable[s] = 1;   // NaN compares false: stays as it is
            st_slot[k] = s;
        }
        {
            const int k0 = has_outlier() ? 1 : 0;          // the outlier component is not part of the Dirichlet (LCA:424-436)
            Philox rng(seed, 0u, draw_epoch, ST_WEIGHTS);
            std::vector<double> g(K - k0 + 1);
            double sum = 0.0;
            for (int k = k0; k < K; ++k) { g[k - k0] = rng.gamma((double)(float)Nrow[3 * slot[k]]); sum += g[k - k0]; }
            g[K - k0] = rng.gamma(alpha); sum += g[K - k0];
            for (int k = k0; k < K; ++k) st_w[k] = (float)((g[k - k0] / sum) * (1.0 - outlier_weight));
            if (k0) st_w[0] = (float)outlier_weight;
        }
        timers[T_MISC] += now_s() - t0; t0 = now_s();
        dev_draw = false;
        if (dev) {      // sample_distribution for all 3K distributions + the hand-over to the sweep kernels, on the device (asynchronous)
            if (mdev) { if (W.mult_draw(W.ctx, draw_epoch, K, has_outlier() ? 1 : 0, st_lr, st_w)) return wfail("mult_draw"); }
            else if (W.niw_draw(W.ctx, draw_epoch, K, st_slot, st_lr, st_w)) return wfail("niw_draw");
            dev_draw = true;
            timers[T_SAMPLE] += now_s() - t0;
        }
        return 0;
    }

    // ---------------------------------------------------------------- steps 5 + 6
    int update_all_with_reset() {
        double t0 = now_s();
        const double *pk = nullptr; const uint8_t *bad = nullptr;
        const bool dev = use_dev();
        bool mdev_marg = false;
        const double *dev_small = nullptr;
        std::vector<int32_t> dev_slots;
        if (dev) {      // statistics + all 3K posteriors and factorisations in one stream-ordered sequence, one wait
            dev_slots.assign(slot.begin(), slot.end());
            // the merge proposals of this step can only involve clusters whose gate is open NOW (splits close gates, they open none):
            // their pooled log-determinants are launched with the posteriors and are ready when check_and_merge asks
            if (W.niw_pairs_ahead && opt_draw_ahead) {
                std::vector<int32_t> ai, aj;
                for (int i = (has_outlier() ? 1 : 0); i < K && ai.size() <= 1024; ++i) {
                    if (!splittable[slot[i]]) continue;
                    for (int j = i + 1; j < K; ++j)
                        if (splittable[slot[j]]) { ai.push_back(slot[i]); aj.push_back(slot[j]); }
                }
                if (ai.size() > 1024) { ai.clear(); aj.clear(); }
                if (W.niw_pairs_ahead(W.ctx, ai.data(), aj.data(), (int)ai.size())) return wfail("niw_pairs_ahead");
            }
            // the epoch of the next parameter draws: the worker launches them right behind the posteriors (they run while this thread
            // decides splits and merges) and uses them if the cluster -> slot map is still this one when sample_clusters asks
            if (W.step_master_device(W.ctx, next_epoch(), dev_slots.data(), opt_draw_ahead ? draw_epoch + 1 : 0u, &bad, &dev_small)) return wfail("step_master_device");
        }
        else {
            // Multinomial with the device master: the log-marginals of the 3K distributions and of the pooled statistics of every pair
            // of clusters whose gate is open NOW (this step can only close gates) ride behind the statistics kernels
            mdev_marg = use_mult_dev() && W.mult_pairs_ahead && W.mult_marginals;
            if (mdev_marg) {
                mp_i.clear(); mp_j.clear(); mp_valid = false;
                for (int i = (has_outlier() ? 1 : 0); i < K && mp_i.size() <= 8192; ++i) {
                    if (!splittable[slot[i]]) continue;
                    for (int j = i + 1; j < K; ++j)
                        if (splittable[slot[j]]) { mp_i.push_back(i); mp_j.push_back(j); }
                }
                if (mp_i.size() > 8192) { mp_i.clear(); mp_j.clear(); }
                if (W.mult_pairs_ahead(W.ctx, has_outlier() ? 1 : 0, mp_i.data(), mp_j.data(), (int)mp_i.size())) return wfail("mult_pairs_ahead");
            }
            mult_pending = nullptr;                 // (the pass rewrites the pinned block; every row is replaced by what it returns)
            if (W.step_stats(W.ctx, next_epoch(), &pk, &bad)) return wfail("step_stats");
        }
        t_stats_back = now_s();
        helper.cancel();                       // a pre-wake still pending is late: drop it
        timers[T_STATS_WAIT] += t_stats_back - t0; t0 = t_stats_back;
        std::vector<int> ks(K);
        int nbad = 0;
        for (int k = 0; k < K; ++k) {
            ks[k] = k;
            if (bad[k]) { splittable[slot[k]] = 0; reset_hist(slot[k]); ++nbad; }     // reset_bad_clusters! (LCA:501-516)
        }
        bad_total += nbad; bad_steps += nbad ? 1 : 0;
        if (dev) { if (int rc = apply_device_posteriors(dev_slots, dev_small)) return rc; dev_state = true; }
        else {
            const double *nl = nullptr, *pl = nullptr;
            int np = 0;
            if (mdev_marg && W.mult_marginals(W.ctx, K, &nl, &pl, &np) == 0 && np == (int)mp_i.size()) {
                ingest_mult_marginals(pk, nl);
                mp_L.assign(pl, pl + np);
                mp_valid = true;
            } else {
                // (the worker may have launched the draws ahead and be delivering the rows behind them: they are read HERE, so wait for them)
                if (W.mult_rows_wait && W.mult_rows_wait(W.ctx)) return wfail("mult_rows_wait");
                ingest(pk, ks); host_dense = true; host_rows = true;
            }
            dev_state = false; mult_rows_current = true;
        }
        timers[T_POSTERIOR] += now_s() - t0;
        return 0;
    }
    int update_subset(const std::vector<int> &ks) {
        if (ks.empty()) return 0;
        double t0 = now_s();
        std::vector<int64_t> idx(ks.size());
        for (size_t i = 0; i < ks.size(); ++i) idx[i] = ks[i] + 1;
        const double *pk = nullptr;
        if (use_dev() && dev_state) {
            if (W.stats_device(W.ctx, idx.data(), (int)idx.size())) return wfail("stats_device");
            timers[T_STATS_WAIT] += now_s() - t0; t0 = now_s();
            if (int rc = ingest_device(ks)) return rc;
            timers[T_POSTERIOR] += now_s() - t0;
            return 0;
        }
        if (int rc = pull_state()) return rc;
        if (W.stats(W.ctx, idx.data(), (int)idx.size(), &pk)) return wfail("stats");
        timers[T_STATS_WAIT] += now_s() - t0; t0 = now_s();
        ingest(pk, ks);
        dev_state = false; mult_rows_current = false;      // (a subset pass: the worker's buffer holds these rows only)
        timers[T_POSTERIOR] += now_s() - t0;
        return 0;
    }
    int update_all_plain() {
        double t0 = now_s();
        const double *pk = nullptr;
        std::vector<int> ks(K);
        for (int k = 0; k < K; ++k) ks[k] = k;
        if (use_dev()) {
            if (W.stats_device(W.ctx, nullptr, 0)) return wfail("stats_device");
            timers[T_STATS_WAIT] += now_s() - t0; t0 = now_s();
            if (int rc = ingest_device(ks)) return rc;
            dev_state = true;
            timers[T_POSTERIOR] += now_s() - t0;
            return 0;
        }
        mult_pending = nullptr;
        if (W.stats(W.ctx, nullptr, 0, &pk)) return wfail("stats");
        timers[T_STATS_WAIT] += now_s() - t0; t0 = now_s();
        ingest(pk, ks);
        host_dense = true; host_rows = true; dev_state = false; mult_rows_current = true;
        timers[T_POSTERIOR] += now_s() - t0;
        return 0;
    }

    // ---------------------------------------------------------------- step 7a: check_and_split! (LCA:345-382)
    double split_log_hr(int k) {
        int sg;
        const int s = slot[k];
        const double nl = Nl(s), nr = Nr(s), nc = nl + nr;
        return log(alpha) + lgamma_r(nl, &sg) + L[3 * s + 1] + lgamma_r(nr, &sg) + L[3 * s + 2] - (lgamma_r(nc, &sg) + L[3 * s]);
    }
    bool split_eligible(int k) {
        const int s = slot[k];
        if (has_outlier() && k == 0) return false;                       // LCA:348-350
        return splittable[s] && Nc(s) > 1 && Nl(s) > 0 && Nr(s) > 0;     // LCA:351, :323
    }
    int check_and_split(bool final, std::vector<int> &touched) {
        touched.clear();
        if (final) return 0;
        split_epoch += 1;
        std::vector<int> acc;
        for (int k = 0; k < K; ++k) {
            if (!split_eligible(k)) continue;
            const double lhr = split_log_hr(k);
            const double u = Philox(seed, (uint32_t)k, split_epoch, ST_SPLIT).uniform();
            if (lhr > log(u)) acc.push_back(k);
        }
        if (acc.empty()) return 0;
        if (kind == DPMMH_PRIOR_MULT) { if (int rc = pull_rows()) return rc; }      // the right halves move to new slots below
        mult_rows_current = false; mp_valid = false;
        const int K0 = K;
        std::vector<int64_t> idx, nidx;
        for (size_t a = 0; a < acc.size(); ++a) {
            const int i = acc[a], j = K0 + (int)a;
            const int sj = alloc_slot();
            if (sj < 0) return -1;
            const int si = slot[i];
            // split_cluster_local! (LCA:280-291): the old cluster keeps its LEFT sub-cluster as cluster, the new one takes the
            // RIGHT; both get fresh sub-clusters (statistics re-computed by the subset pass that follows, LCA:668)
            memcpy(prow(sj, 0), prow(si, 1), sizeof(double) * stride);
            std::fill(prow(sj, 1), prow(sj, 1) + stride, 0.0);
            std::fill(prow(si, 1), prow(si, 1) + stride, 0.0);
            copy_row_post(3 * sj, 3 * si + 2); copy_row_post(3 * sj + 1, 3 * sj); copy_row_post(3 * sj + 2, 3 * sj);
            copy_row_post(3 * si, 3 * si + 1); copy_row_post(3 * si + 2, 3 * si);
            for (int s : {si, sj}) {
                splittable[s] = 0; reset_hist(s);
                points_count[s] = (int64_t)llrint(Nrow[3 * s]);
            }
            slot.push_back(sj);
            idx.push_back(i + 1); nidx.push_back(j + 1);
        }
        K = (int)slot.size();
        if (W.set_num_clusters(W.ctx, K)) return wfail("set_num_clusters");
        if (W.split(W.ctx, idx.data(), nidx.data(), (int)idx.size(), next_epoch())) return wfail("split");
        for (auto v : idx) touched.push_back((int)v - 1);
        for (auto v : nidx) touched.push_back((int)v - 1);
        if (hook) {
            const double t0 = now_s();
            std::vector<int64_t> all(touched.begin(), touched.end());
            for (auto &v : all) v += 1;
            if (hook(hook_user, all.data(), (int)all.size())) return fail("split hook failed");
            timers[T_HOOK] += now_s() - t0;
        }
        return 0;
    }

    // ---------------------------------------------------------------- step 7c: check_and_merge! (LCA:385-413, SA:21-38)
    // SA:28-30.  The terms that depend on ONE cluster (or on alpha alone) are looked up: K (K - 1) / 2 pairs share 2 K + 3 log-gamma values
    // (merge_cache, refreshed by check_and_merge); the sum keeps the order it had with every term evaluated in place, so the ratio is the
    // same double.
    std::vector<double> mc_lgN, mc_lgNa;        // lgamma(N_k), lgamma(N_k + alpha / 2) by cluster index
    double mc_head = 0.0;                       // -log(alpha) + lgamma(alpha) - 2 lgamma(alpha / 2)
    void merge_cache() {
        int sg;
        const double a = alpha;
        mc_head = -log(a) + lgamma_r(a, &sg) - 2 * lgamma_r(0.5 * a, &sg);
        mc_lgN.resize(K); mc_lgNa.resize(K);
        for (int k = 0; k < K; ++k) { const double N = Nc(slot[k]); mc_lgN[k] = lgamma_r(N, &sg); mc_lgNa[k] = lgamma_r(N + 0.5 * a, &sg); }
    }
    bool mc_exact = false;                      // dpmmh_debug_merge_log_hr: every ratio in full
    double merge_log_hr(int i, int j, double Lp) {
        int sg;
        const double a = alpha, Np = Nc(slot[i]) + Nc(slot[j]);
        // lgamma(Np) - lgamma(Np + alpha) <= 0: when the other terms alone are far below what any uniform accepts (check_and_merge refuses below
        // -38; the 22 in between is head-room for the rounding of sums of order 1e6), the two log-gammas of the pair -- all that is left of the
        // K (K - 1) / 2 x 3 of the formula, and 20 us per step at K = 32 -- are not taken.  The value returned is then a bound, not the ratio.
        if (!mc_exact) {
            const double rest = mc_head + mc_lgNa[i] - mc_lgN[i] - mc_lgN[j] + mc_lgNa[j] + Lp - L[3 * slot[i]] - L[3 * slot[j]];
            if (rest < -60.0) return rest;
        }
        return mc_head + lgamma_r(Np, &sg) - lgamma_r(Np + a, &sg) +
               mc_lgNa[i] - mc_lgN[i] - mc_lgN[j] + mc_lgNa[j] + Lp - L[3 * slot[i]] - L[3 * slot[j]];
    }
    // log_marginal_likelihood of the pooled statistics of clusters (i, j) under cluster i's prior (SA:22-27)
    double pooled_marginal(int i, int j, std::vector<double> &sc) {
        const int si = slot[i], sj = slot[j];
        const double *rows[4] = {prow(si, 0), prow(si, 1), prow(sj, 0), prow(sj, 1)};
        if (kind == DPMMH_PRIOR_NIW) {
            const NiwPrior &pr = niw_of(si);
            const size_t DD = (size_t)D * D;
            if (sc.size() < DD + (size_t)D) sc.resize(DD + (size_t)D);
            double *P = sc.data(), *mm = P + DD;
            double N = 0.0;
            for (auto *r : rows) N += r[0];
            if (N == 0.0) return niw_marginal(pr, pr.kappa, pr.nu, pr.logdet_psi, 0.0);
            const double k0 = pr.kappa, v0 = pr.nu, k1 = k0 + N, v1 = v0 + N;
            for (int a = 0; a < D; ++a) mm[a] = (pr.m[a] * k0 + (rows[0][1 + a] + rows[1][1 + a] + rows[2][1 + a] + rows[3][1 + a])) / k1;
            const double *m0 = pr.m.data();
            for (int a = 0; a < D; ++a) {
                const size_t t0 = 1 + (size_t)D + (size_t)a * (a + 1) / 2;
                const double *r0 = rows[0] + t0, *r1 = rows[1] + t0, *r2 = rows[2] + t0, *r3 = rows[3] + t0;
                const double *pa = pr.psi_lo.data() + (size_t)a * (a + 1) / 2;
                double *Pa = P + (size_t)a * D;
                const double km0a = k0 * m0[a], kma = k1 * mm[a];
#pragma omp simd
                for (int b = 0; b <= a; ++b) {
                    const double sab = r0[b] + r1[b] + r2[b] + r3[b];
                    Pa[b] = ((v0 * pa[b] + km0a * m0[b] - kma * mm[b] + sab) / v1) * v1;       // lower triangle
                }
            }
            const double ld = dpmmh::logdet_spd_inplace(P, D) - D * log(v1);
            return niw_marginal(pr, k1, v1, ld, N);
        }
        const MultPrior &pr = mult_of(si);
        double N = 0.0;
        for (auto *r : rows) N += r[0];
        if (N == 0.0) return 0.0;
        // multinomial_prior.jl:34-39 for the pooled posterior alpha + sum of the four rows, in one pass: the prior's two sums are
        // constants, the pooled alpha' is never stored (K (K - 1) / 2 pairs of D lookups each are the master's largest item at D = 1000)
        const dpmmh::LgammaTable &lg = dpmmh::LgammaTable::get();
        const float *al = pr.alpha.data();
        const double *r0 = rows[0] + 1, *r1 = rows[1] + 1, *r2 = rows[2] + 1, *r3 = rows[3] + 1;
        double s1 = 0.0, acc = 0.0;
        for (int d = 0; d < D; ++d) {
            const float a1 = al[d] + (float)(r0[d] + r1[d] + r2[d] + r3[d]);
            s1 += (double)a1;
            acc += lg(a1);
        }
        int sg;
        return lgamma_r(pr.alpha_sum, &sg) - lgamma_r(s1, &sg) + (acc - pr.lg_alpha_sum);
    }
    void merge_candidates(std::vector<std::pair<int, int>> &pairs) {
        pairs.clear();
        for (int i = (has_outlier() ? 1 : 0); i < K; ++i) {             // LCA:390-392
            if (!(splittable[slot[i]] && Nc(slot[i]) > 0)) continue;
            for (int j = i + 1; j < K; ++j)
                if (splittable[slot[j]] && Nc(slot[j]) > 0) pairs.emplace_back(i, j);
        }
    }
    void merge_ratios(const std::vector<std::pair<int, int>> &pairs, std::vector<double> &lhr) {
        lhr.resize(pairs.size());
        merge_cache();
        if (dev_pairs_ok) {
            // pooled scale matrices and their log-determinants on the device, from the rows it keeps per slot
            const int n = (int)pairs.size();
            std::vector<int32_t> si(n), sj(n);
            for (int p = 0; p < n; ++p) { si[p] = slot[pairs[p].first]; sj[p] = slot[pairs[p].second]; }
            const double *sm = nullptr;
            if (W.niw_pairs(W.ctx, si.data(), sj.data(), n, &sm) == 0) {
                const NiwPrior &pr = niw[0];
                auto one = [&](int p) {
                    const double *o = sm + (size_t)p * kDevScalars;
                    double Lp;
                    if (o[0] == 0.0) Lp = niw_marginal(pr, pr.kappa, pr.nu, pr.logdet_psi, 0.0);
                    else if (f32_quirk) Lp = niw_marginal(pr, o[1], o[2], o[3] - D * log(o[2]), o[0]);
                    else Lp = dpmmh::niw_log_marginal_lmg(D, pr.kappa, pr.nu, pr.logdet_psi, pr.lmg0[0], o[1], o[2], o[3] - D * log(o[2]), o[0], o[4]);
                    lhr[p] = merge_log_hr(pairs[p].first, pairs[p].second, Lp);
                };
                if (f32_quirk) Pool::get().run(n, nthreads, [&](int p, int) { one(p); });
                else for (int p = 0; p < n; ++p) one(p);
                return;
            }
            wfail("niw_pairs");      // fall through to the host path (rows are fetched below by the caller's pull_rows)
            dev_pairs_ok = false;
            pull_rows();
        }
        std::vector<std::vector<double>> scratch(std::max(1, nthreads));
        if (kind == DPMMH_PRIOR_MULT && mp_valid && use_mult_dev()) {      // pooled log-marginals the worker computed behind the statistics; the rest on the host
            // both lists are in lexicographic order (the candidates are those of the pairs computed ahead whose gates are still open): one walk
            std::vector<int> todo;
            size_t q = 0;
            for (int p = 0; p < (int)pairs.size(); ++p) {
                const int i = pairs[p].first, j = pairs[p].second;
                while (q < mp_i.size() && (mp_i[q] < i || (mp_i[q] == i && mp_j[q] < j))) ++q;
                if (q < mp_i.size() && mp_i[q] == i && mp_j[q] == j) lhr[p] = merge_log_hr(i, j, mp_L[q]);
                else todo.push_back(p);
            }
            if (!todo.empty() && pull_rows()) { for (int p : todo) lhr[p] = -INFINITY; return; }      // (no rows: no merge of those pairs; the failure is recorded)
            if (!todo.empty()) Pool::get().run((int)todo.size(), nthreads, [&](int t, int th) {
                const int p = todo[t];
                lhr[p] = merge_log_hr(pairs[p].first, pairs[p].second, pooled_marginal(pairs[p].first, pairs[p].second, scratch[th]));
            });
            return;
        }
        Pool::get().run((int)pairs.size(), nthreads, [&](int p, int th) {
            lhr[p] = merge_log_hr(pairs[p].first, pairs[p].second, pooled_marginal(pairs[p].first, pairs[p].second, scratch[th]));
        });
    }
    int check_and_merge(bool final) {
        std::vector<std::pair<int, int>> pairs;
        merge_candidates(pairs);
        if (pairs.empty()) return 0;
        dev_pairs_ok = use_dev() && dev_state && !host_rows;     // the device has the rows: it forms and factorises the pooled matrices
        // else: pooled statistics are formed from the host's rows (Multinomial with the worker's pair marginals: only for pairs it did not evaluate
        // and for accepted merges)
        if (!dev_pairs_ok && !(kind == DPMMH_PRIOR_MULT && mp_valid && use_mult_dev())) { if (int rc = pull_rows()) return rc; }
        merge_epoch += 1;
        double t0 = now_s();
        std::vector<double> lhr;
        merge_ratios(pairs, lhr);
        timers[T_MERGE_PAIRS] += now_s() - t0;
        // the reference walks the pairs in lexicographic order and updates the state in between: a merged pair leaves both
        // clusters non-splittable, so later pairs that involve either are never evaluated
        std::vector<uint8_t> used(K, 0);
        std::vector<int64_t> idx, nidx;
        std::vector<double> sc;
        for (size_t p = 0; p < pairs.size(); ++p) {
            const int i = pairs[p].first, j = pairs[p].second;
            if (used[i] || used[j]) continue;
            // (the uniform is keyed by the pair, not drawn from a running stream, and is at least 2^-54: a ratio below log(2^-54) = -37.4 --
            // two distinct clusters sit at -1e4 and beyond -- is refused whatever it would have been, and log(0.1) is above that too)
            if (lhr[p] < -38.0) continue;
            const double u = Philox(seed, (uint32_t)(i * 65536 + j), merge_epoch, ST_MERGE).uniform();
            if (!((lhr[p] > log(u)) || (final && lhr[p] > log(0.1)))) continue;
            if (int rc = pull_state()) return rc;      // an accepted merge rebuilds posteriors on the host (no-op when they are current)
            used[i] = used[j] = 1;
            const int si = slot[i], sj = slot[j];
            // merge_clusters_to_splittable (SA:12-18): left := old cluster i, right := old cluster j, cluster := their sum
            double *li = prow(si, 0), *ri = prow(si, 1), *lj = prow(sj, 0), *rj = prow(sj, 1);
            for (int64_t e = 0; e < stride; ++e) { li[e] += ri[e]; ri[e] = lj[e] + rj[e]; }
            std::fill(lj, lj + stride, 0.0); std::fill(rj, rj + stride, 0.0);
            copy_row_post(3 * si + 1, 3 * si);
            copy_row_post(3 * si + 2, 3 * sj);
            refresh_row(si, 0, sc);
            splittable[si] = 0; reset_hist(si);
            points_count[si] += points_count[sj];
            points_count[sj] = 0;
            Nrow[3 * sj] = 0.0;
            splittable[sj] = 0;
            idx.push_back(i + 1); nidx.push_back(j + 1);
        }
        if (idx.empty()) return 0;
        dev_state = false; mult_rows_current = false; mp_valid = false;  // merged slots were rebuilt on the host: the next draws come from there
        if (W.merge(W.ctx, idx.data(), nidx.data(), (int)idx.size())) return wfail("merge");
        return 0;
    }

    // ---------------------------------------------------------------- step 8: remove_empty_clusters! (LCA:457-471)
    int remove_empty() {
        std::vector<int64_t> pc(K);
        bool any = false;
        for (int k = 0; k < K; ++k) {
            const bool keep = points_count[slot[k]] > 0 || (has_outlier() && k == 0) || (has_outlier() && k == 1 && K == 2);
            pc[k] = keep ? std::max<int64_t>(points_count[slot[k]], 1) : 0;
            any |= !keep;
        }
        if (!any) return 0;
        if (kind == DPMMH_PRIOR_MULT) { if (int rc = pull_rows()) return rc; }      // (pending rows are in the cluster order that ends here)
        if (W.remove_empty(W.ctx, pc.data(), K)) return wfail("remove_empty");
        mult_rows_current = false; mp_valid = false;   // (the worker's rows are in the old cluster order)
        std::vector<int> ns;
        for (int k = 0; k < K; ++k) {
            if (pc[k] > 0) ns.push_back(slot[k]);
            else slot_used[slot[k]] = 0;
        }
        slot.swap(ns);
        K = (int)slot.size();
        if (W.set_num_clusters(W.ctx, K)) return wfail("set_num_clusters");
        return 0;
    }

    // ---------------------------------------------------------------- the sweep (LCA:658-673)
    int group_step(bool no_more_splits, bool final) {
        if (int rc = sample_clusters()) return rc;                                   // 1
        double t0 = now_s();
        if (!dev_draw && W.commit_params(W.ctx, K)) return wfail("commit_params");   // 2 (device draws are handed over where they are made)
        timers[T_COMMIT] += now_s() - t0; t0 = now_s();
        if (W.sweep(W.ctx, next_epoch(), (final || hard) ? 1 : 0)) return wfail("sweep");   // 3 + 4 (asynchronous); LCA:661
        const double t_launch = now_s();
        start_noise(t_launch);                                                       // the host works while the GPU sweeps
        timers[T_SWEEP_LAUNCH] += now_s() - t0;
        if (int rc = update_all_with_reset()) return rc;                             // 5 + 6
        {
            // launch-to-statistics time of this step (before the posterior work): the next step's pre-wake prediction
            const double w = t_stats_back - t_launch;
            wait_ema = wait_ema > 0.0 ? 0.5 * wait_ema + 0.5 * w : w;
        }
        if (!no_more_splits) {                                                       // 7
            t0 = now_s();
            std::vector<int> touched;
            if (int rc = check_and_split(final, touched)) return rc;
            timers[T_SPLIT] += now_s() - t0;
            if (int rc = update_subset(touched)) return rc;
            t0 = now_s();
            if (int rc = check_and_merge(final)) return rc;
            timers[T_MERGE] += now_s() - t0;
        }
        t0 = now_s();
        const int rc = remove_empty();                                               // 8
        timers[T_REMOVE] += now_s() - t0;
        return rc;
    }

    int set_K(int Knew) {
        if (Knew < 1) return fail("K < 1");
        for (int s : slot) slot_used[s] = 0;
        slot.clear();
        if (ensure_slots(Knew)) return -1;
        for (int k = 0; k < Knew; ++k) { const int s = alloc_slot(); if (s < 0) return -1; slot.push_back(s); }
        K = Knew;
        return 0;
    }
};

// ===================================================================================================== C ABI
#define HAPI extern "C" __attribute__((visibility("default")))

HAPI int dpmmh_abi_version(void) { return DPMMH_ABI_VERSION; }

HAPI int dpmmh_model_create(dpmmh_model **out, int prior_kind, int D, double alpha, int64_t n_total, uint64_t seed, int burnout, int nthreads) {
    if (!out) return -1;
    *out = nullptr;
    if ((prior_kind != DPMMH_PRIOR_NIW && prior_kind != DPMMH_PRIOR_MULT) || D < 1 || burnout < 1 || !(alpha > 0)) return -1;
    dpmmh_model *m = new dpmmh_model();
    m->kind = prior_kind; m->D = D; m->alpha = (double)(float)alpha; m->n_total = n_total; m->seed = seed;
    m->burnout = burnout; m->hist_len = burnout + 5;
    m->nthreads = std::max(1, nthreads);
    m->T = prior_kind == DPMMH_PRIOR_NIW ? (int64_t)D * (D + 1) / 2 : 0;
    m->stride = 1 + (int64_t)D + m->T;
    Pool::get().set_spin_us(150);
    *out = m;
    return 0;
}

HAPI void dpmmh_model_destroy(dpmmh_model *m) {
    if (!m) return;
    m->wait_noise();
    delete m;
}

HAPI const char *dpmmh_model_last_error(const dpmmh_model *m) { return m ? m->err.c_str() : "null model"; }

HAPI int dpmmh_model_set_prior_niw(dpmmh_model *m, int which, double kappa, const double *mean, double nu, const double *psi) {
    if (!m || which < 0 || which > 1 || !mean || !psi) return -1;
    if (m->kind != DPMMH_PRIOR_NIW) return m->fail("model was created for another prior");
    NiwPrior &p = m->niw[which];
    const int D = m->D;
    p.kappa = (double)(float)kappa; p.nu = (double)(float)nu;          // Float32 in the reference (niw.jl:6-11)
    p.m.assign(mean, mean + D); p.psi.assign(psi, psi + (size_t)D * D);
    p.psi_lo.resize((size_t)D * (D + 1) / 2);
    dpmmh::pack_sym_lower(D, psi, p.psi_lo.data());
    std::vector<double> tmp(p.psi);
    p.logdet_psi = dpmmh::logdet_spd_inplace(tmp.data(), D);
    p.lmg0[0] = dpmmh::log_multivariate_gamma(p.nu / 2.0, D, false);
    p.lmg0[1] = dpmmh::log_multivariate_gamma(p.nu / 2.0, D, true);
    p.set = true;
    return 0;
}

HAPI int dpmmh_model_set_prior_mult(dpmmh_model *m, int which, const float *alpha) {
    if (!m || which < 0 || which > 1 || !alpha) return -1;
    if (m->kind != DPMMH_PRIOR_MULT) return m->fail("model was created for another prior");
    m->mult[which].alpha.assign(alpha, alpha + m->D);
    {
        const dpmmh::LgammaTable &lg = dpmmh::LgammaTable::get();
        double s0 = 0.0, l0 = 0.0;
        for (int d = 0; d < m->D; ++d) { s0 += (double)alpha[d]; l0 += lg(alpha[d]); }
        m->mult[which].alpha_sum = s0; m->mult[which].lg_alpha_sum = l0;
    }
    m->mult[which].set = true;
    return 0;
}

HAPI int dpmmh_model_set_outlier(dpmmh_model *m, double w) {
    if (!m) return -1;
    if (!(w >= 0 && w < 1)) return m->fail("outlier weight must be in [0, 1)");
    m->outlier_weight = w;
    return 0;
}

HAPI int dpmmh_model_set_option(dpmmh_model *m, int option, double value) {
    if (!m) return -1;
    switch (option) {
        case DPMMH_OPT_HARD_CLUSTERING: m->hard = value != 0; return 0;
        case DPMMH_OPT_F32_QUIRK: m->f32_quirk = value != 0; return 0;
        case DPMMH_OPT_THREADS: m->nthreads = std::max(1, (int)value); return 0;
        case DPMMH_OPT_SHARE_WORK: if (value != 0) return m->fail("DPMMH_OPT_SHARE_WORK: owner-computes sharing of the master's work is not implemented"); m->share_work = false; return 0;
        case DPMMH_OPT_SPIN_US: Pool::get().set_spin_us((int)value); return 0;
        case DPMMH_OPT_PREWAKE: m->prewake = value != 0; return 0;
        case DPMMH_OPT_DRAW_AHEAD: m->opt_draw_ahead = value != 0; return 0;
        case DPMMH_OPT_DEVICE_MASTER: m->opt_dev_master = value < 0 ? -1 : (value != 0); if (!m->opt_dev_master) { if (m->pull_state()) return -1; m->dev_state = false; } return 0;
        case DPMMH_OPT_NUMA_NODE: {
            const int node = (int)value;
            Pool::get().set_numa_node(node);
            m->helper.set_numa_node(node);
            return 0;
        }
        default: return m->fail("unknown option");
    }
}

HAPI int dpmmh_model_bind_worker(dpmmh_model *m, const dpmmh_worker *w) {
    if (!m || !w) return -1;
    if (!w->params_staging || !w->commit_params || !w->set_num_clusters || !w->sweep || !w->step_stats || !w->stats || !w->split ||
        !w->merge || !w->remove_empty || !w->reset_sublabels || !w->init_labels)
        return m->fail("worker table has null entries");
    m->W = *w;
    m->bound = true;
    m->st_slots = 0;
    return 0;
}

HAPI int dpmmh_model_set_split_hook(dpmmh_model *m, dpmmh_split_hook hook, void *user) {
    if (!m) return -1;
    m->hook = hook; m->hook_user = user;
    return 0;
}

static int check_ready(dpmmh_model *m) {
    if (!m) return -1;
    if (!m->bound) return m->fail("no worker bound");
    if (m->kind == DPMMH_PRIOR_NIW ? !m->niw[0].set : !m->mult[0].set) return m->fail("prior not set");
    if (m->has_outlier() && (m->kind == DPMMH_PRIOR_NIW ? !m->niw[1].set : !m->mult[1].set)) return m->fail("outlier prior not set");
    return 0;
}

HAPI int dpmmh_model_init_first_clusters(dpmmh_model *m, int init_clusters) {
    if (int rc = check_ready(m)) return rc;
    if (init_clusters < 1) return m->fail("init_clusters < 1");
    const int out = m->has_outlier() ? 1 : 0;
    if (int rc = m->set_K(init_clusters + out)) return rc;
    // labels = rand(1:init_clusters) (.+ 1 with an outlier component), sub-labels = rand(1:2)   (dp-parallel-sampling.jl:49-50)
    if (m->W.init_labels(m->W.ctx, init_clusters, 1 + out, m->next_epoch())) return m->wfail("init_labels");
    if (m->W.set_num_clusters(m->W.ctx, m->K)) return m->wfail("set_num_clusters");
    if (m->W.reset_sublabels(m->W.ctx, nullptr, 0, m->next_epoch())) return m->wfail("reset_sublabels");   // split_first_cluster_worker!
    if (int rc = m->update_all_plain()) return rc;
    if (m->hook) {                                                         // dp-parallel-sampling.jl:70-75
        std::vector<int64_t> all(m->K);
        for (int k = 0; k < m->K; ++k) all[k] = k + 1;
        if (m->hook(m->hook_user, all.data(), m->K)) return m->fail("split hook failed");
        if (int rc = m->update_all_plain()) return rc;
    }
    return m->sample_clusters();
}

HAPI int dpmmh_model_start_from_labels(dpmmh_model *m, int K) {
    if (int rc = check_ready(m)) return rc;
    if (int rc = m->set_K(K)) return rc;
    if (m->W.set_num_clusters(m->W.ctx, m->K)) return m->wfail("set_num_clusters");
    if (int rc = m->update_all_plain()) return rc;
    return m->sample_clusters();
}

HAPI int dpmmh_group_step(dpmmh_model *m, int no_more_splits, int final) {
    if (int rc = check_ready(m)) return rc;
    if (m->K < 1) return m->fail("no clusters: call dpmmh_model_init_first_clusters first");
    return m->group_step(no_more_splits != 0, final != 0);
}

HAPI int dpmmh_sample_clusters(dpmmh_model *m) {
    if (int rc = check_ready(m)) return rc;
    return m->sample_clusters();
}

HAPI int dpmmh_update_suff_stats_posterior(dpmmh_model *m, const int64_t *clusters, int n) {
    if (int rc = check_ready(m)) return rc;
    if (!clusters) return m->update_all_plain();
    std::vector<int> ks;
    for (int i = 0; i < n; ++i) {
        if (clusters[i] < 1 || clusters[i] > m->K) return m->fail("cluster id out of range");
        ks.push_back((int)clusters[i] - 1);
    }
    return m->update_subset(ks);
}

HAPI double dpmmh_log_posterior(dpmmh_model *m) {
    if (!m || m->K < 1) return NAN;
    int sg;
    double lp = lgamma_r(m->alpha, &sg) - lgamma_r((double)m->n_total + m->alpha, &sg);
    for (int k = 0; k < m->K; ++k) {
        const int s = m->slot[k];
        const double N = m->Nrow[3 * s];
        if (N == 0.0) continue;
        lp += m->L[3 * s] + log(m->alpha) + lgamma_r(N, &sg);
    }
    return lp;
}

HAPI const char *dpmmh_timer_names(void) { return kTimerNames; }

// ----------------------------------------------------------------------------------------------------- state access
namespace {
template <typename T>
int64_t emit(void *out, int64_t cap, const std::vector<T> &v) {
    const int64_t bytes = (int64_t)(v.size() * sizeof(T));
    if (out) {
        if (cap < bytes) return -3;
        memcpy(out, v.data(), bytes);
    }
    return bytes;
}
}  // namespace

HAPI int64_t dpmmh_model_get(dpmmh_model *m, const char *field, void *out, int64_t cap) {
    if (!m || !field) return -1;
    const std::string f(field);
    const int K = m->K, D = m->D;
    const size_t DD = (size_t)D * D;
    if (f == "K") return emit<int64_t>(out, cap, {(int64_t)K});
    // counters[6]: bit 0 = the device master is on but the NEXT draws of this chain come from the host (the step that just ended accepted a
    // merge and rebuilt the merged slots there; Multinomial: a split, merge or removal left the worker without the current rows): host and
    // device draws use different streams, so a resumed run has to know
    if (f == "counters") return emit<int64_t>(out, cap, {(int64_t)m->epoch, (int64_t)m->draw_epoch, (int64_t)m->split_epoch, (int64_t)m->merge_epoch, m->bad_total, m->bad_steps,
                                                         (int64_t)(((m->kind == DPMMH_PRIOR_NIW && m->use_dev() && !m->dev_state) ||
                                                                    (m->kind == DPMMH_PRIOR_MULT && m->use_mult_dev() && !m->mult_rows_current)) ? 1 : 0), 0});
    if (f == "timers") return emit<double>(out, cap, std::vector<double>(m->timers, m->timers + 16));
    auto rows_d = [&](const std::vector<double> &src, size_t w) {
        std::vector<double> v((size_t)3 * K * w);
        for (int k = 0; k < K; ++k) for (int r = 0; r < 3; ++r) memcpy(&v[((size_t)3 * k + r) * w], &src[((size_t)3 * m->slot[k] + r) * w], sizeof(double) * w);
        return v;
    };
    auto rows_f = [&](const float *src, size_t w) {
        std::vector<float> v((size_t)3 * K * w);
        for (int k = 0; k < K; ++k) for (int r = 0; r < 3; ++r) memcpy(&v[((size_t)3 * k + r) * w], &src[((size_t)3 * m->slot[k] + r) * w], sizeof(float) * w);
        return v;
    };
    if (f == "N") return emit(out, cap, rows_d(m->Nrow, 1));
    if (f == "kappa") return emit(out, cap, rows_d(m->kappa, 1));
    if (f == "nu") return emit(out, cap, rows_d(m->nu, 1));
    if (f == "logdet_psi") return emit(out, cap, rows_d(m->ldpsi, 1));
    if (f == "log_marginal") return emit(out, cap, rows_d(m->L, 1));
    if (f == "packed" || f == "sums" || f == "S") { if (m->pull_rows()) return -1; }      // the device may hold the only current copy
    if (f == "m" || f == "U" || f == "alpha_post") { if (m->pull_state()) return -1; }
    if (m->kind == DPMMH_PRIOR_NIW && m->dev_draw && (f == "mu" || f == "R" || f == "logdet")) {
        // the current draws were made on the device: fetch them (cluster order)
        std::vector<float> mu((size_t)3 * K * D), R((size_t)3 * K * DD), ld((size_t)3 * K);
        if (m->W.niw_draws(m->W.ctx, K, mu.data(), R.data(), ld.data())) return m->wfail("niw_draws");
        return emit(out, cap, f == "mu" ? mu : (f == "R" ? R : ld));
    }
    if (f == "packed") {
        std::vector<double> v((size_t)2 * K * m->stride);
        for (int k = 0; k < K; ++k) memcpy(&v[(size_t)2 * k * m->stride], m->prow(m->slot[k], 0), sizeof(double) * 2 * m->stride);
        return emit(out, cap, v);
    }
    if (f == "sums" || f == "S") {
        const bool isS = f == "S";
        if (isS && m->kind != DPMMH_PRIOR_NIW) return m->fail("no S for this prior");
        const size_t w = isS ? DD : (size_t)D;
        const int64_t bytes = (int64_t)((size_t)3 * K * w * sizeof(double));
        if (!out) return bytes;
        if (cap < bytes) return -3;
        double *o = (double *)out;
        for (int k = 0; k < K; ++k) {
            const double *l = m->prow(m->slot[k], 0), *r = m->prow(m->slot[k], 1);
            for (int ww = 0; ww < 3; ++ww) {
                const double cl = (ww != 2) ? 1.0 : 0.0, cr = (ww != 1) ? 1.0 : 0.0;
                double *dst = o + ((size_t)3 * k + ww) * w;
                if (!isS) { for (int d = 0; d < D; ++d) dst[d] = cl * l[1 + d] + cr * r[1 + d]; continue; }
                for (int a = 0; a < D; ++a)
                    for (int b = 0; b <= a; ++b) {
                        const size_t t = 1 + (size_t)D + (size_t)a * (a + 1) / 2 + b;
                        dst[(size_t)a * D + b] = dst[(size_t)b * D + a] = cl * l[t] + cr * r[t];
                    }
            }
        }
        return bytes;
    }
    if (m->kind == DPMMH_PRIOR_NIW) {
        if (f == "m") return emit(out, cap, rows_d(m->mean, D));
        if (f == "U") {          // the factor is stored as L = U' (row-major lower): hand out U (upper) with nu psi = U U'
            std::vector<double> v = rows_d(m->U, DD), t(DD);
            for (size_t r3 = 0; r3 < (size_t)3 * K; ++r3) {
                double *Mx = v.data() + r3 * DD;
                for (int a = 0; a < D; ++a) for (int b = 0; b < D; ++b) t[(size_t)a * D + b] = b >= a ? Mx[(size_t)b * D + a] : 0.0;
                memcpy(Mx, t.data(), sizeof(double) * DD);
            }
            return emit(out, cap, v);
        }
        if (m->st_slots > 0) {
            if (f == "mu") return emit(out, cap, rows_f(m->st_mu, D));
            if (f == "R") {          // staging holds the packed upper triangle: hand out the full square
                std::vector<float> v((size_t)3 * K * DD, 0.f);
                for (int k = 0; k < K; ++k) for (int r3 = 0; r3 < 3; ++r3) {
                    const float *src = m->st_mat + ((size_t)3 * m->slot[k] + r3) * m->T;
                    float *dst = v.data() + ((size_t)3 * k + r3) * DD;
                    for (int r = 0; r < D; ++r) memcpy(dst + (size_t)r * D + r, src + (size_t)r * D - (size_t)r * (r - 1) / 2, sizeof(float) * (D - r));
                }
                return emit(out, cap, v);
            }
            if (f == "logdet") return emit(out, cap, rows_f(m->st_logdet, 1));
        }
    } else {
        if (f == "alpha_post") return emit(out, cap, rows_f(m->apost.data(), D));
        if (f == "logp" && m->dev_draw) {      // the current draws were made on the device: fetch them (cluster order)
            std::vector<float> lp((size_t)3 * K * D);
            if (m->W.mult_draws(m->W.ctx, K, lp.data())) return m->wfail("mult_draws");
            return emit(out, cap, lp);
        }
        if (f == "logp" && m->st_slots > 0) return emit(out, cap, rows_f(m->st_mat, D));
    }
    if (f == "lr_weights" && m->st_slots > 0) return emit(out, cap, std::vector<float>(m->st_lr, m->st_lr + 2 * K));
    if (f == "weights" && m->st_slots > 0) return emit(out, cap, std::vector<float>(m->st_w, m->st_w + K));
    if (f == "splittable") { std::vector<uint8_t> v(K); for (int k = 0; k < K; ++k) v[k] = m->splittable[m->slot[k]]; return emit(out, cap, v); }
    if (f == "points_count") { std::vector<int64_t> v(K); for (int k = 0; k < K; ++k) v[k] = m->points_count[m->slot[k]]; return emit(out, cap, v); }
    if (f == "hist") {
        std::vector<float> v((size_t)K * m->hist_len);
        for (int k = 0; k < K; ++k) memcpy(&v[(size_t)k * m->hist_len], &m->hist[(size_t)m->slot[k] * m->hist_len], sizeof(float) * m->hist_len);
        return emit(out, cap, v);
    }
    return m->fail("unknown field " + f);
}

HAPI int dpmmh_model_set(dpmmh_model *m, const char *field, const void *in, int64_t bytes) {
    if (!m || !field || !in) return -1;
    const std::string f(field);
    const int D = m->D;
    const size_t DD = (size_t)D * D;
    if (f == "K") {
        if (bytes != 8) return m->fail("K: expected one int64");
        if (!m->bound) return m->fail("bind a worker before restoring state");
        return m->set_K((int)*(const int64_t *)in);
    }
    const int K = m->K;
    auto need = [&](int64_t want) { if (bytes != want) { m->fail("field " + f + ": wrong size"); return false; } return true; };
    if (f == "counters") {
        if (!need(64)) return -1;
        const int64_t *c = (const int64_t *)in;
        m->epoch = (uint32_t)c[0]; m->draw_epoch = (uint32_t)c[1]; m->split_epoch = (uint32_t)c[2]; m->merge_epoch = (uint32_t)c[3];
        if (c[6] & 1) m->mult_rows_current = false;
        if ((c[6] & 1) && m->dev_state) {       // saved right after an accepted merge: the running chain made its next draws on the host
            if (m->pull_state()) return -1;
            m->dev_state = false;
        }
        return 0;
    }
    if (K < 1) return m->fail("set K first");
    if (f == "packed") {
        if (!need((int64_t)sizeof(double) * 2 * K * m->stride)) return -1;
        std::vector<int> ks(K);
        for (int k = 0; k < K; ++k) ks[k] = k;
        m->ingest((const double *)in, ks);
        m->host_dense = true; m->host_rows = true; m->dev_state = false; m->dev_draw = false; m->mult_rows_current = false; m->mp_valid = false;
        if (m->use_mult_dev()) {
            if (m->W.mult_put_rows(m->W.ctx, (const double *)in, K)) return m->wfail("mult_put_rows");
            m->mult_rows_current = true;
            // (the running chain's log-marginals came from the device: a resumed one takes them from there too)
            const double *nl = nullptr, *pl = nullptr;
            int np = 0;
            if (m->W.mult_pairs_ahead && m->W.mult_marginals && m->W.mult_pairs_ahead(m->W.ctx, m->has_outlier() ? 1 : 0, nullptr, nullptr, 0) == 0 &&
                m->W.mult_marginals(m->W.ctx, K, &nl, &pl, &np) == 0)
                m->ingest_mult_marginals((const double *)in, nl);
            m->mult_pending = nullptr; m->host_rows = true;      // (the rows went in through ingest above; `in` is the caller's)
        }
        if (m->use_dev()) {      // the device gets the same rows, so that the next draws come from where a running chain makes them
            if (m->W.niw_put_rows(m->W.ctx, (const double *)in, K)) return m->wfail("niw_put_rows");
            if (int rc = m->ingest_device(ks)) return rc;
            m->dev_state = true; m->host_dense = true; m->host_rows = true;
        }
        return 0;
    }
    auto rows_f = [&](float *dst, size_t w) {
        const float *src = (const float *)in;
        for (int k = 0; k < K; ++k) for (int r = 0; r < 3; ++r) memcpy(&dst[((size_t)3 * m->slot[k] + r) * w], &src[((size_t)3 * k + r) * w], sizeof(float) * w);
    };
    if (f == "mu" && m->kind == DPMMH_PRIOR_NIW) { if (!need(4LL * 3 * K * D)) return -1; rows_f(m->st_mu, D); return 0; }
    if (f == "R" && m->kind == DPMMH_PRIOR_NIW) {
        if (!need((int64_t)(4 * 3 * K * DD))) return -1;
        const float *src = (const float *)in;
        for (int k = 0; k < K; ++k) for (int r3 = 0; r3 < 3; ++r3) {
            float *dst = m->st_mat + ((size_t)3 * m->slot[k] + r3) * m->T;
            const float *sq = src + ((size_t)3 * k + r3) * DD;
            for (int r = 0; r < D; ++r) memcpy(dst + (size_t)r * D - (size_t)r * (r - 1) / 2, sq + (size_t)r * D + r, sizeof(float) * (D - r));
        }
        return 0;
    }
    if (f == "logdet" && m->kind == DPMMH_PRIOR_NIW) { if (!need(4LL * 3 * K)) return -1; rows_f(m->st_logdet, 1); return 0; }
    if (f == "logp" && m->kind == DPMMH_PRIOR_MULT) { if (!need(4LL * 3 * K * D)) return -1; rows_f(m->st_mat, D); return 0; }
    if (f == "lr_weights") { if (!need(8LL * K)) return -1; memcpy(m->st_lr, in, bytes); return 0; }
    if (f == "weights") { if (!need(4LL * K)) return -1; memcpy(m->st_w, in, bytes); for (int k = 0; k < K; ++k) m->st_slot[k] = m->slot[k]; return 0; }
    if (f == "splittable") { if (!need(K)) return -1; for (int k = 0; k < K; ++k) m->splittable[m->slot[k]] = ((const uint8_t *)in)[k]; return 0; }
    if (f == "points_count") { if (!need(8LL * K)) return -1; for (int k = 0; k < K; ++k) m->points_count[m->slot[k]] = ((const int64_t *)in)[k]; return 0; }
    if (f == "hist") {
        if (!need(4LL * K * m->hist_len)) return -1;
        for (int k = 0; k < K; ++k) memcpy(&m->hist[(size_t)m->slot[k] * m->hist_len], (const float *)in + (size_t)k * m->hist_len, sizeof(float) * m->hist_len);
        return 0;
    }
    return m->fail("unknown / read-only field " + f);
}

HAPI int dpmmh_debug_split_log_hr(dpmmh_model *m, double *out) {
    if (!m || !out) return -1;
    for (int k = 0; k < m->K; ++k) out[k] = m->split_eligible(k) ? m->split_log_hr(k) : NAN;
    return 0;
}

HAPI int dpmmh_debug_merge_log_hr(dpmmh_model *m, double *out) {
    if (!m || !out) return -1;
    const int K = m->K;
    for (int e = 0; e < K * K; ++e) out[e] = NAN;
    std::vector<std::pair<int, int>> pairs;
    std::vector<double> lhr;
    m->merge_candidates(pairs);
    // as check_and_merge: the pooled statistics come from the device when it holds the current rows, else from the host's (pulled if stale)
    m->dev_pairs_ok = m->use_dev() && m->dev_state && !m->host_rows;
    if (!m->dev_pairs_ok) if (int rc = m->pull_rows()) return rc;
    m->mc_exact = true;
    m->merge_ratios(pairs, lhr);
    m->mc_exact = false;
    for (size_t p = 0; p < pairs.size(); ++p) out[(size_t)pairs[p].first * K + pairs[p].second] = lhr[p];
    return 0;
}
