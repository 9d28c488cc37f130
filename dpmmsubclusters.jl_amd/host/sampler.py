"""Master-side sampler: the host half of one restricted-Gibbs sweep.

The master's work between the workers' calls -- parameter draws, posteriors, split / merge Metropolis steps, compaction --
runs in NATIVE code (host/csrc/dpmm_model.cpp behind include/dpmm_host.h; one `dpmmh_group_step` call is one sweep and
drives the GPU worker through the addresses of the libdpmmhip.so entry points).  This module is the thin host-language
layer around it: the iteration loop with the reference's schedule flags (run_model), the smart-split initialisation hook,
evaluation against a ground truth, and read access to the cluster state.  Reference functions covered by the engine
(paths relative to the reference checkout):

    group_step                          src/local_clusters_actions.jl:658-673
    sample_clusters!                    src/local_clusters_actions.jl:417-437
    sample_cluster_params               src/shared_actions.jl:41-66   (burn-in gate :51-63)
    update_suff_stats_posterior!        src/local_clusters_actions.jl:206-254
    reset_bad_clusters!                 src/local_clusters_actions.jl:501-516
    check_and_split! / should_split_local! / split_cluster_local!   :345-382, :318-343, :280-291
    check_and_merge! / should_merge! / merge_clusters!              :385-413, shared_actions.jl:21-38, :308-315
    remove_empty_clusters!              src/local_clusters_actions.jl:457-471
    init_first_clusters!                src/dp-parallel-sampling.jl:62-78
    run_model / calculate_posterior     src/dp-parallel-sampling.jl:336-404, :458-470

Cluster k occupies rows 3k (cluster), 3k+1 (left), 3k+2 (right) of every per-distribution array.
In a multi-GPU run every rank runs the same engine on identical all-reduced statistics with identical
counter-based randomness, so no parameter broadcast is needed (the reference's broadcast_cluster_params,
:518-549, becomes a local hand-over on every rank).
"""
import time

import numpy as np

from . import engine as _engine

# schedule constants that `fit` cannot change (src/global_params.jl:10-11)
ARGMAX_SAMPLE_STOP = 5
SPLIT_STOP = 5


class LocalComm:
    """Single process, single GPU: the statistics come straight off the device."""
    rank, world = 0, 1

    def gather_labels(self, worker):
        return worker.get_labels()

    def reduce_counts(self, table):
        return table

    def reduce_f64(self, arr, op="sum"):
        return np.asarray(arr, np.float64)

    def attach(self, worker):
        """Single rank: nothing to attach."""


def nmi_vi_from_contingency(C):
    """NMI (Clustering.jl `mutualinfo(a, b, normed=true)` = 2 I / (H_a + H_b)) and VI (`varinfo` = H_a + H_b - 2 I)
    from a contingency table -- what run_model logs per iteration (src/dp-parallel-sampling.jl:370-377)."""
    C = np.asarray(C, np.float64)
    N = C.sum()
    if N == 0:
        return 0.0, 0.0
    pij = C / N
    pi = pij.sum(1); pj = pij.sum(0)
    nz = pij > 0
    mi = float((pij[nz] * np.log(pij[nz] / (pi[:, None] * pj[None, :])[nz])).sum())
    hi = float(-(pi[pi > 0] * np.log(pi[pi > 0])).sum()); hj = float(-(pj[pj > 0] * np.log(pj[pj > 0])).sum())
    nmi = 2 * mi / (hi + hj) if (hi + hj) > 0 else 1.0
    return nmi, hi + hj - 2 * mi


class DPMMSampler:
    def __init__(self, worker, prior, alpha, n_total, seed, burnout=20, max_clusters=np.inf, comm=None,
                 argmax_sample_stop=ARGMAX_SAMPLE_STOP, split_stop=SPLIT_STOP, nthreads=None):
        from .. import binding
        from . import native as _native
        self.wk = worker
        self.prior = prior
        self.alpha = float(np.float32(alpha))
        self.n_total = int(n_total)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.burnout = int(burnout)
        self.max_clusters = max_clusters
        self.comm = comm or LocalComm()
        self.argmax_sample_stop = argmax_sample_stop
        self.split_stop = split_stop
        self.nthreads = int(nthreads or _native.default_threads())
        self.hard_clustering = False   # global_params.jl:8: argmax label assignment in every sweep
        self.smart_splits = False      # global_params.jl:44 / fit(...; smart_splits): Gaussian prior only
        self.max_split_iter = 20       # global_params.jl:15
        self.outlier_weight = 0.0      # outlier_mod: constant weight of an extra, never-splitting component at index 1
        self.outlier_prior = None      # outlier_hyper_params
        self.f32_quirk = False         # utils.jl:66-72 Float32 accumulator of log_multivariate_gamma (reference-compat switch)
        self.vi_history = []
        self.model = _engine.Model(prior.kind, prior.dim, self.alpha, self.n_total, self.seed, self.burnout, self.nthreads)
        self._set_prior(0, prior)
        if isinstance(worker, binding.Worker):
            self.comm.attach(worker)                       # multi-GPU: the RCCL all-reduce lives inside libdpmmhip.so
            table, keep = _engine.native_worker_table(worker, self.comm.rank, self.comm.world)
        else:
            table, keep = _engine.python_worker_table(worker, self.comm)
        self.model.bind_worker(table, keep)
        self._configured = False

    # ------------------------------------------------------------------ configuration
    def _set_prior(self, which, prior):
        if prior.kind == 0:
            self.model.set_prior_niw(which, prior.kappa, prior.m, prior.nu, prior.psi)
        else:
            self.model.set_prior_mult(which, prior.alpha)

    def _configure(self):
        """Push the attributes callers set after construction (fit / dp_parallel keyword arguments) to the engine."""
        m = self.model
        m.set_option(_engine.OPT_HARD_CLUSTERING, 1.0 if self.hard_clustering else 0.0)
        m.set_option(_engine.OPT_F32_QUIRK, 1.0 if self.f32_quirk else 0.0)
        m.set_option(_engine.OPT_THREADS, self.nthreads)
        if hasattr(self.wk, "numa_node"):        # two-socket hosts: the master's threads next to the GPU's host memory
            m.set_option(_engine.OPT_NUMA_NODE, self.wk.numa_node())
        if self.outlier_weight > 0:
            self._set_prior(1, self.outlier_prior)
            m.set_outlier(self.outlier_weight)
        else:
            m.set_outlier(0.0)
        m.set_split_hook(self._smart_hook if self.smart_splits else None)
        self._configured = True

    def _smart_hook(self, clusters_1based):
        for k in clusters_1based:
            self.smart_cluster_init(int(k) - 1)

    # ------------------------------------------------------------------ state (read access, cluster order)
    @property
    def K(self):
        return self.model.K

    @property
    def N(self):
        return self.model.get("N").reshape(-1, 3)

    @property
    def sums(self):
        return self.model.get("sums").reshape(self.K, 3, self.prior.dim)

    @property
    def S(self):
        return self.model.get("S").reshape(self.K, 3, self.prior.dim, self.prior.dim) if self.prior.kind == 0 else None

    @property
    def post(self):
        if self.prior.kind == 0:
            return {k: self.model.get(k) for k in ("kappa", "nu", "m", "U", "logdet_psi")}
        return dict(alpha=self.model.get("alpha_post"))

    @property
    def params(self):
        if self.prior.kind == 0:
            return dict(mu=self.model.get("mu"), R=self.model.get("R"), logdet=self.model.get("logdet"))
        return dict(logp=self.model.get("logp"))

    weights = property(lambda self: self.model.get("weights"))
    lr_weights = property(lambda self: self.model.get("lr_weights"))
    splittable = property(lambda self: self.model.get("splittable").astype(bool))
    hist = property(lambda self: self.model.get("hist"))
    points_count = property(lambda self: self.model.get("points_count"))
    log_marginals = property(lambda self: self.model.get("log_marginal").reshape(-1, 3))

    @property
    def timers(self):
        return self.model.timers()

    @property
    def epoch(self):
        return int(self.model.get("counters")[0])

    # ------------------------------------------------------------------ smart splits
    def smart_cluster_init(self, k):
        """smart_cluster_init!(group, cluster_num) (local_clusters_actions.jl:555-627), k 0-based.  The direction is taken
        exactly as the reference takes it -- `F.vectors[argmax(F.values), :]`, i.e. a ROW of the eigenvector matrix of the
        cluster covariance, and the two seeds are `percentile(t, 0.10)` / `percentile(t, 0.90)` in StatsBase's 0..100
        convention (the 0.001 and 0.009 quantiles) -- see DESIGN.md for why these two quirks are kept.  Every rank computes the
        direction from the same all-reduced statistics, so the projections agree across shards."""
        if self.prior.kind != 0:
            return
        N = self.N[k, 0]
        if not (N > 0):
            return
        XXT = self.S[k, 0] / N
        mu = self.sums[k, 0] / N
        M = XXT - np.outer(mu, mu)
        vals, vecs = np.linalg.eigh(M)                      # Hermitian path of Julia's eigen(): ascending eigenvalues
        v1 = np.ascontiguousarray(vecs[int(np.argmax(vals)), :])
        t = self.wk.smart_project(k + 1, v1, mu)
        lo, hi = np.inf, -np.inf
        if len(t) > 1:                                       # `length(transformed_pts) > 1`, else the worker returns nothing
            lo, hi = np.quantile(t, 0.10 * 0.01), np.quantile(t, 0.90 * 0.01)
        lo = float(self.comm.reduce_f64([lo], "min")[0]); hi = float(self.comm.reduce_f64([hi], "max")[0])
        if not np.isfinite(lo) and not np.isfinite(hi) and lo > hi:
            return                                           # no worker had more than one point of this cluster
        it, converged = 0, False
        while it < self.max_split_iter and not converged:
            s = self.comm.reduce_f64(self.wk.smart_kmeans_iter(k + 1, lo, hi), "sum")
            with np.errstate(invalid="ignore", divide="ignore"):
                new_lo, new_hi = float(np.float64(s[0]) / np.float64(s[1])), float(np.float64(s[2]) / np.float64(s[3]))
            if new_lo == lo and new_hi == hi:
                converged = True
            else:
                lo, hi = new_lo, new_hi
            it += 1
        self.wk.smart_assign(k + 1, lo, hi)

    # ------------------------------------------------------------------ the sweep
    def group_step(self, no_more_splits, final):
        """group_step (local_clusters_actions.jl:658-673): one native call."""
        if not self._configured:
            self._configure()
        self.model.group_step(no_more_splits, final)

    def sample_clusters(self):
        self.model.sample_clusters()

    def update_suff_stats_posterior(self, ks=None):
        """ks: 0-based cluster ids (None = all)."""
        self.model.update_suff_stats_posterior(None if ks is None else np.asarray(ks, np.int64) + 1)

    def init_first_clusters(self, init_clusters):
        """init_model_from_data labels (dp-parallel-sampling.jl:49-50) + init_first_clusters! (:62-78)."""
        self._configure()
        self.model.init_first_clusters(int(init_clusters))

    def start_from_labels(self, labels, sub_labels, K):
        """Resume / benchmark entry: adopt given (local shard) labels instead of random ones."""
        self._configure()
        self.wk.set_labels(labels, sub_labels)
        self.model.start_from_labels(int(K))

    def log_posterior(self):
        """calculate_posterior (dp-parallel-sampling.jl:458-470)."""
        return self.model.log_posterior()

    def run_model(self, iterations, first_iter=1, verbose=False, gt=None, on_iteration=None):
        """run_model (dp-parallel-sampling.jl:336-404): returns iter_count, nmi_history, likelihood_history, cluster_count_history."""
        iter_count, nmi_hist, lik_hist, k_hist = [], [], [], []
        self.vi_history = []
        if not self._configured:
            self._configure()
        if gt is not None:
            # ground truth of this shard goes to the GPU once; per iteration only a K x n_gt table comes back
            ids, inv = np.unique(np.asarray(gt), return_inverse=True)
            lo = getattr(self.wk, "first_index", 0)
            self.wk.set_ground_truth_range(inv[lo:lo + self.wk.n], len(ids))
        K = self.K
        for i in range(first_iter, iterations + 1):
            final = i >= iterations - self.argmax_sample_stop
            no_more_splits = (i >= iterations - self.split_stop) or (K >= self.max_clusters)
            t0 = time.perf_counter()
            self.model.group_step(no_more_splits, final)
            dt = time.perf_counter() - t0
            iter_count.append(dt)
            K = self.K
            k_hist.append(K)
            if gt is not None:
                nmi, vi = nmi_vi_from_contingency(self.comm.reduce_counts(self.wk.contingency(K)))
                nmi_hist.append(nmi); self.vi_history.append(vi)
            else:
                nmi_hist.append("no gt"); self.vi_history.append("no gt")
            if verbose:
                lik_hist.append(self.log_posterior())
                if self.comm.rank == 0:
                    print(f"Iteration: {i} || Clusters count: {K} || Log posterior: {lik_hist[-1]} || NMI score: {nmi_hist[-1]}"
                          f" || Iter Time:{dt} || Total time:{sum(iter_count)}")
            else:
                lik_hist.append(1)
            if on_iteration is not None:
                on_iteration(i, self)
        return iter_count, nmi_hist, lik_hist, k_hist
