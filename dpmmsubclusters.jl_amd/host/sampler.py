"""Master-side sampler: the host half of one restricted-Gibbs sweep.

Restates, over struct-of-arrays cluster state and the GPU `Worker` (one per rank), the
master-process functions of the reference (paths relative to the reference checkout):

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
In a multi-GPU run every rank executes this file redundantly on identical all-reduced statistics
with identical counter-based randomness, so no parameter broadcast is needed (the reference's
broadcast_cluster_params, :518-549, becomes a local H2D copy on every rank).
"""
import time

import numpy as np
from scipy.special import gammaln

# schedule constants that `fit` cannot change (src/global_params.jl:10-11)
ARGMAX_SAMPLE_STOP = 5
SPLIT_STOP = 5


class LocalComm:
    """Single process, single GPU: the statistics come straight off the device."""
    rank, world = 0, 1

    def reduce_stats(self, worker, idx):
        return worker.suffstats_packed(idx)

    def gather_labels(self, worker):
        return worker.get_labels()

    def reduce_counts(self, table):
        return table

    def reduce_f64(self, arr, op="sum"):
        return np.asarray(arr, np.float64)

    def broadcast(self, arr, src=0):
        return arr


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
        self.nthreads = nthreads
        self.hard_clustering = False   # global_params.jl:8: argmax label assignment in every sweep
        self.smart_splits = False      # global_params.jl:44 / fit(...; smart_splits): Gaussian prior only
        self.max_split_iter = 20       # global_params.jl:15
        self.outlier_weight = 0.0      # outlier_mod: constant weight of an extra, never-splitting component at index 1
        self.outlier_prior = None      # outlier_hyper_params
        self._outlier_params = None
        self.rng = np.random.Generator(np.random.Philox(key=self.seed))
        self.epoch = 0          # device-side randomised calls (stream-unique)
        self.draw_epoch = 1 << 20   # host parameter draws: separate, predictable counter (noise is pre-generated)
        self.K = 0
        self.timers = {}
        self._noise_job = None
        self._executor = None
        self._gen = 0             # bumped whenever posteriors / statistics / K change: validates the log-marginal cache
        self._L_cache = None
        # Leader mode (multi-rank only): rank 0 alone runs the heavy host maths (posterior factorisations, parameter
        # draws, merge log-marginals) with all host threads and broadcasts the small results; every rank still takes
        # the same Metropolis decisions from the same numbers.  Chosen when redundant execution would leave each rank
        # with only a few host threads (e.g. a container CPU quota); DPMM_LEADER_MODE=0/1 overrides.
        import os
        from . import native as _native
        env = os.environ.get("DPMM_LEADER_MODE")
        self.leader_mode = self.comm.world > 1 and (env == "1" or (env is None and (nthreads or _native.default_threads()) < 8))
        self.is_leader = self.comm.rank == 0
        if self.leader_mode and self.is_leader and nthreads is None:
            self.nthreads = max(1, min(32, _native._cpu_budget() - 2))

    # ------------------------------------------------------------------ small helpers
    def _next_epoch(self):
        self.epoch += 1
        return self.epoch

    def _tic(self, name, t0):
        self.timers[name] = self.timers.get(name, 0.0) + (time.perf_counter() - t0)

    def _dirichlet(self, a):
        g = self.rng.standard_gamma(np.asarray(a, np.float64))
        return g / g.sum(-1, keepdims=True)

    def _rows(self, ks):
        ks = np.asarray(ks, np.int64)
        return (3 * ks[:, None] + np.arange(3)[None, :]).ravel()

    def _alloc(self, K):
        self._touch()
        D = self.prior.dim
        self.K = K
        st = self.prior.empty_stats(3 * K)
        self.N = st["N"].reshape(K, 3); self.sums = st["sums"].reshape(K, 3, D)
        self.S = st["S"].reshape(K, 3, D, D) if st["S"] is not None else None
        self.post = None
        self.params = None
        self.lr_weights = np.full((K, 2), 0.5, np.float32)
        self.weights = np.full(K, 1.0 / K, np.float32)
        self.splittable = np.zeros(K, bool)
        self.hist = np.full((K, self.burnout + 5), -np.inf, np.float32)
        self.points_count = np.zeros(K, np.int64)

    def _set_post_rows(self, rows, post):
        self._touch()
        if self.post is None:
            self.post = {k: np.array(v) for k, v in post.items()}
            return
        for k, v in post.items():
            self.post[k][rows] = v

    def _stats_flat(self):
        K, D = self.K, self.prior.dim
        return self.N.reshape(3 * K), self.sums.reshape(3 * K, D), (self.S.reshape(3 * K, D, D) if self.S is not None else None)

    # ------------------------------------------------------------------ noise prefetch (overlaps the GPU sweep)
    def _start_noise(self):
        """Generate the standard-normal part of the NEXT parameter draws on a helper thread while the GPU sweeps."""
        if not hasattr(self.prior, "draw_noise") or (self.leader_mode and not self.is_leader):
            return
        if self._executor is None:       # one persistent helper thread (creating a thread per sweep costs ~0.1 ms)
            from concurrent.futures import ThreadPoolExecutor
            self._executor = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dpmm-noise")
        rows = 3 * (self.K + 4)          # head-room for clusters born from splits
        epoch = self.draw_epoch + 1
        job = dict(epoch=epoch, rows=rows)
        job["future"] = self._executor.submit(self.prior.draw_noise, rows, self.seed, epoch, nthreads=self.nthreads)
        self._noise_job = job

    def _touch(self):
        self._gen += 1

    def _log_marginal_all(self):
        """log_marginal of all 3K statistic sets, cached until the posteriors change (three callers per sweep)."""
        if self._L_cache is not None and self._L_cache[0] == self._gen and len(self._L_cache[1]) == 3 * self.K:
            return self._L_cache[1]
        L = self.prior.log_marginal(self.post, self.N.reshape(3 * self.K))
        self._L_cache = (self._gen, L)
        return L

    def _take_noise(self, epoch, rows):
        job, self._noise_job = self._noise_job, None
        if job is None:
            return None
        out = job["future"].result()
        if job["epoch"] != epoch or job["rows"] < rows:
            return None
        return out

    # ------------------------------------------------------------------ statistics (step 5)
    def update_suff_stats_posterior(self, ks=None):
        """ks: 0-based cluster ids (None = all).  One GPU statistics pass + (multi-GPU) one all-reduce."""
        if ks is not None and len(ks) == 0:
            return
        t0 = time.perf_counter()
        self._touch()
        idx = None if ks is None else (np.asarray(ks, np.int64) + 1)
        packed = self.comm.reduce_stats(self.wk, idx)
        self._tic("stats_gpu", t0)
        t0 = time.perf_counter()
        sel = np.arange(self.K) if ks is None else np.asarray(ks, np.int64)
        if hasattr(self.prior, "update_from_packed"):
            if self.post is None or len(self.post["kappa"]) != 3 * self.K:
                self.post = self.prior.empty_post(3 * self.K)
            if not self.leader_mode or self.is_leader:
                self.prior.update_from_packed(packed, None if ks is None else sel, self.N, self.sums, self.S, self.post,
                                              nthreads=self.nthreads)
            self._sync_small()
            self.points_count[sel] = np.rint(self.N[sel, 0]).astype(np.int64)
            if self.outlier_weight > 0:
                self.points_count[0] = self.n_total          # create_outlier_local_cluster: never refreshed, never empty
            self._tic("posterior_host", t0)
            return
        un = self.wk.unpack(packed, self.K)
        N, sums = un[0], un[1]
        S = un[2] if len(un) > 2 else None
        self.N[sel] = N[sel]; self.sums[sel] = sums[sel]
        if S is not None:
            self.S[sel] = S[sel]
        self.points_count[sel] = np.rint(N[sel, 0]).astype(np.int64)
        if self.outlier_weight > 0:
            self.points_count[0] = self.n_total
        rows = self._rows(sel)
        Nf, sf, Sf = self._stats_flat()
        post = self.prior.posterior(Nf[rows], sf[rows], Sf[rows] if Sf is not None else None, nthreads=self.nthreads)
        if ks is None:
            self.post = post
        else:
            self._set_post_rows(rows, post)
        self._tic("posterior_host", t0)

    def _sync_small(self):
        """Leader mode: the per-distribution scalars every rank needs for its (identical) decisions."""
        if not self.leader_mode or self.post is None or "kappa" not in self.post:
            return
        K = self.K
        buf = np.empty((4, 3 * K))
        if self.is_leader:
            buf[0] = self.N.reshape(3 * K); buf[1] = self.post["kappa"]; buf[2] = self.post["nu"]; buf[3] = self.post["logdet_psi"]
        self.comm.broadcast(buf)
        if not self.is_leader:
            self.N[:] = buf[0].reshape(K, 3)
            self.post["kappa"][:] = buf[1]; self.post["nu"][:] = buf[2]; self.post["logdet_psi"][:] = buf[3]

    def _sync_params(self):
        if not self.leader_mode:
            return
        for k in sorted(self.params):
            self.comm.broadcast(self.params[k])

    # ------------------------------------------------------------------ step 1
    def sample_clusters(self):
        t0 = time.perf_counter()
        K = self.K
        self.draw_epoch += 1
        noise = self._take_noise(self.draw_epoch, 3 * K)
        if self.leader_mode and not self.is_leader:
            self.params = self.prior.empty_params(3 * K)
        elif noise is not None:
            self.params = self.prior.sample(self.post, self.seed, self.draw_epoch, np.arange(3 * K), nthreads=self.nthreads, noise=noise)
        else:
            self.params = self.prior.sample(self.post, self.seed, self.draw_epoch, np.arange(3 * K), nthreads=self.nthreads)
        if self.outlier_weight > 0 and self._outlier_params is not None:
            for key, val in self._outlier_params.items():   # sample_clusters! skips index 1 (local_clusters_actions.jl:424-427)
                self.params[key][0:3] = val
        self._sync_params()
        self._tic("sample_params_host", t0)
        t0 = time.perf_counter()
        half = self.alpha / 2
        self.lr_weights = self._dirichlet(self.N[:, 1:3] + half).astype(np.float32)
        L = self._log_marginal_all().reshape(K, 3)
        b = self.burnout
        self.hist[:, : b - 1] = self.hist[:, 1:b]
        with np.errstate(invalid="ignore", over="ignore"):
            self.hist[:, b - 1] = (L[:, 1] + L[:, 2]).astype(np.float32)
            now = (self.hist[:, :b].astype(np.float64) * (1.0 / (b - 0.1))).sum(1)
            gate = (now != -np.inf) & ((now - self.hist[:, b - 1]) < 1e-2)
        self.splittable |= gate
        if self.outlier_weight > 0:
            ow = self.outlier_weight                          # local_clusters_actions.jl:431-436
            self.lr_weights[0] = 0.5
            self.splittable[0] = False
            self.hist[0] = -np.inf
            w = self._dirichlet(np.concatenate([self.N[1:, 0], [self.alpha]]))
            self.weights = np.concatenate([[ow], w[:K - 1] * (1.0 - ow)]).astype(np.float32)
        else:
            w = self._dirichlet(np.concatenate([self.N[:, 0], [self.alpha]]))
            self.weights = w[:K].astype(np.float32)
        self._tic("host_misc", t0)

    # ------------------------------------------------------------------ step 6
    def reset_bad_clusters(self):
        bad = np.flatnonzero((self.N[:, 1] == 0) | (self.N[:, 2] == 0))
        if len(bad) == 0:
            return
        self.hist[bad] = -np.inf
        self.splittable[bad] = False
        self.wk.reset_sublabels(bad + 1, self._next_epoch())
        self.update_suff_stats_posterior(bad)

    def update_stats_and_reset_bad(self):
        """Steps 5 + 6 of group_step (local_clusters_actions.jl:665-666) with ONE statistics pass: the sub-cluster
        occupancies (the N of the l / r statistics) come from the sort histogram first, clusters with an empty
        sub-cluster get their sub-labels re-drawn (reset_bad_clusters!, :501-516), and the statistics / posteriors
        are then computed once over the final labelling.  Same end state as the reference's full pass + subset pass:
        a sub-label reset changes neither cluster-level statistics nor any other cluster."""
        t0 = time.perf_counter()
        counts = self.comm.reduce_counts(self.wk.bin_counts())
        bad = np.flatnonzero((counts[:, 0] == 0) | (counts[:, 1] == 0))
        if len(bad):
            self.hist[bad] = -np.inf
            self.splittable[bad] = False
            self.wk.reset_sublabels(bad + 1, self._next_epoch())
        self._tic("bad_reset", t0)
        self.update_suff_stats_posterior()

    # ------------------------------------------------------------------ smart splits
    def smart_cluster_init(self, k):
        """smart_cluster_init!(group, cluster_num) (local_clusters_actions.jl:555-627), k 0-based.  The direction is taken
        exactly as the reference takes it -- `F.vectors[argmax(F.values), :]`, i.e. a ROW of the eigenvector matrix of the
        cluster covariance, and the two seeds are `percentile(t, 0.10)` / `percentile(t, 0.90)` in StatsBase's 0..100
        convention (the 0.001 and 0.009 quantiles) -- see DESIGN.md for why these two quirks are kept."""
        N = self.N[k, 0]
        if not (N > 0) or self.S is None:
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

    # ------------------------------------------------------------------ step 7a
    def check_and_split(self, final):
        K = self.K
        if final:
            return np.zeros(0, np.int64)
        cand = np.flatnonzero(self.splittable & (self.N[:, 0] > 1) & (self.N[:, 1] > 0) & (self.N[:, 2] > 0))
        if self.outlier_weight > 0:
            cand = cand[cand != 0]                            # local_clusters_actions.jl:348-350
        if len(cand) == 0:
            return np.zeros(0, np.int64)
        L = self._log_marginal_all().reshape(K, 3)
        Nc, Nl, Nr = self.N[cand, 0], self.N[cand, 1], self.N[cand, 2]
        log_hr = np.log(self.alpha) + gammaln(Nl) + L[cand, 1] + gammaln(Nr) + L[cand, 2] - (gammaln(Nc) + L[cand, 0])
        u = self.rng.random(len(cand))
        with np.errstate(divide="ignore"):
            acc = cand[log_hr > np.log(u)]
        if len(acc) == 0:
            return np.zeros(0, np.int64)
        new = K + np.arange(len(acc))
        self._grow(K + len(acc))
        for i, j in zip(acc, new):      # split_cluster_local!: old <- left, new <- right
            ri, rj = 3 * i, 3 * j
            self._copy_row(rj, ri + 2); self._copy_row(rj + 1, rj); self._copy_row(rj + 2, rj)
            self._copy_row(ri, ri + 1); self._copy_row(ri + 1, ri); self._copy_row(ri + 2, ri)
            for k in (i, j):
                self.lr_weights[k] = self._dirichlet([self.alpha / 2, self.alpha / 2]).astype(np.float32)
                self.splittable[k] = False
                self.hist[k] = -np.inf
                self.points_count[k] = int(round(self.N[k, 0]))
        self.wk.set_num_clusters(self.K)
        self.wk.split(acc + 1, new + 1, self._next_epoch())
        touched = np.concatenate([acc, new])
        if self.smart_splits:                                # local_clusters_actions.jl:374-378
            for k in touched:
                self.smart_cluster_init(int(k))
        return touched

    def _grow(self, K2):
        self._touch()
        K, D = self.K, self.prior.dim
        add = K2 - K
        self.N = np.concatenate([self.N, np.zeros((add, 3))])
        self.sums = np.concatenate([self.sums, np.zeros((add, 3, D))])
        if self.S is not None:
            self.S = np.concatenate([self.S, np.zeros((add, 3, D, D))])
        for d in (self.post, self.params):
            for k in list(d.keys()):
                d[k] = np.concatenate([d[k], np.zeros((3 * add,) + d[k].shape[1:], d[k].dtype)])
        self.lr_weights = np.concatenate([self.lr_weights, np.full((add, 2), 0.5, np.float32)])
        self.weights = np.concatenate([self.weights, np.zeros(add, np.float32)])
        self.splittable = np.concatenate([self.splittable, np.zeros(add, bool)])
        self.hist = np.concatenate([self.hist, np.full((add, self.hist.shape[1]), -np.inf, np.float32)])
        self.points_count = np.concatenate([self.points_count, np.zeros(add, np.int64)])
        self.K = K2

    def _copy_row(self, dst, src):
        self._touch()
        """Copy one distribution row (statistics, posterior, drawn parameters)."""
        kd, wd = divmod(dst, 3); ks, ws = divmod(src, 3)
        self.N[kd, wd] = self.N[ks, ws]; self.sums[kd, wd] = self.sums[ks, ws]
        if self.S is not None:
            self.S[kd, wd] = self.S[ks, ws]
        for d in (self.post, self.params):
            for k in d:
                d[k][dst] = d[k][src]

    # ------------------------------------------------------------------ step 7c
    def check_and_merge(self, final):
        K = self.K
        ok = self.splittable & (self.N[:, 0] > 0)
        ids = np.flatnonzero(ok)
        if len(ids) < 2:
            return
        ii, jj = np.triu_indices(len(ids), 1)
        pi, pj = ids[ii], ids[jj]                     # lexicographic (i<j) order
        Nf, sf, Sf = self._stats_flat()
        Lc = self._log_marginal_all().reshape(K, 3)[:, 0]
        t0 = time.perf_counter()
        if not self.leader_mode or self.is_leader:
            Lp = self.prior.log_marginal_pairs(np.stack([3 * pi, 3 * pj], 1), dict(N=Nf, sums=sf, S=Sf), nthreads=self.nthreads)
        else:
            Lp = np.empty(len(pi))
        Lp = self.comm.broadcast(np.ascontiguousarray(Lp, np.float64)) if self.leader_mode else Lp
        self._tic("merge_pairs_host", t0)
        a = self.alpha
        Ni, Nj = self.N[pi, 0], self.N[pj, 0]
        Np = Ni + Nj
        log_hr = (-np.log(a) + gammaln(a) - 2 * gammaln(0.5 * a) + gammaln(Np) - gammaln(Np + a)
                  + gammaln(Ni + 0.5 * a) - gammaln(Ni) - gammaln(Nj) + gammaln(Nj + 0.5 * a) + Lp - Lc[pi] - Lc[pj])
        u = self.rng.random(len(pi))
        with np.errstate(divide="ignore"):
            acc = (log_hr > np.log(u)) | (final & (log_hr > np.log(0.1)))
        used = np.zeros(K, bool)
        m_i, m_j = [], []
        for p in np.flatnonzero(acc):
            i, j = pi[p], pj[p]
            if used[i] or used[j]:
                continue
            used[i] = used[j] = True
            m_i.append(i); m_j.append(j)
        if not m_i:
            return
        for i, j in zip(m_i, m_j):                 # merge_clusters! / merge_clusters_to_splittable
            ri, rj = 3 * i, 3 * j
            Ni_, Nj_ = self.N[i, 0], self.N[j, 0]
            self._copy_row(ri + 1, ri)             # left  := old cluster i
            self._copy_row(ri + 2, rj)             # right := old cluster j
            self.N[i, 0] = Ni_ + Nj_
            self.sums[i, 0] = self.sums[i, 1] + self.sums[i, 2]
            if self.S is not None:
                self.S[i, 0] = self.S[i, 1] + self.S[i, 2]
            Nf, sf, Sf = self._stats_flat()
            if not self.leader_mode or self.is_leader or "kappa" not in self.post:
                self._set_post_rows([ri], self.prior.posterior(Nf[[ri]], sf[[ri]], Sf[[ri]] if Sf is not None else None, nthreads=1))
            self.lr_weights[i] = self._dirichlet([Ni_ + a / 2, Nj_ + a / 2]).astype(np.float32)
            self.splittable[i] = False
            self.hist[i] = -np.inf
            self.points_count[i] += self.points_count[j]
            self.points_count[j] = 0
            self.N[j, 0] = 0
            self.splittable[j] = False
            self._touch()
        self._sync_small()
        self.wk.merge(np.asarray(m_i) + 1, np.asarray(m_j) + 1)

    # ------------------------------------------------------------------ step 8
    def remove_empty_clusters(self):
        keep = self.points_count > 0
        if keep.all():
            return
        self._touch()
        self.wk.remove_empty(self.points_count)
        rows = self._rows(np.flatnonzero(keep))
        self.N = self.N[keep]; self.sums = self.sums[keep]
        if self.S is not None:
            self.S = self.S[keep]
        for d in (self.post, self.params):
            for k in list(d.keys()):
                d[k] = d[k][rows]
        self.lr_weights = self.lr_weights[keep]; self.weights = self.weights[keep]
        self.splittable = self.splittable[keep]; self.hist = self.hist[keep]
        self.points_count = self.points_count[keep]
        self.K = int(keep.sum())
        self.wk.set_num_clusters(self.K)

    # ------------------------------------------------------------------ the sweep
    def group_step(self, no_more_splits, final):
        self.sample_clusters()                                   # 1
        t0 = time.perf_counter()
        self.prior.upload(self.wk, self.params, self.lr_weights, self.weights)   # 2
        self._tic("upload_params", t0)
        t0 = time.perf_counter()
        self.wk.sweep(self._next_epoch(), bool(final or self.hard_clustering))   # 3 + 4 (asynchronous); LCA:661
        self._start_noise()                                      # host works while the GPU sweeps
        self._tic("sweep_launch", t0)
        self.update_stats_and_reset_bad()                        # 5 + 6
        if not no_more_splits:                                   # 7
            t0 = time.perf_counter()
            touched = self.check_and_split(final)
            self._tic("split_host", t0)
            self.update_suff_stats_posterior(touched)
            t0 = time.perf_counter()
            self.check_and_merge(final)
            self._tic("merge_host", t0)
        self.remove_empty_clusters()                             # 8

    def _create_outlier_cluster(self):
        """create_outlier_local_cluster (local_clusters_actions.jl:42-61): statistics of ALL points under the outlier prior,
        one parameter draw shared by the cluster and both sub-clusters, never re-drawn."""
        op = self.outlier_prior
        Nt = np.array([self.N[1:, 0].sum()])
        st = self.sums[1:, 0].sum(0)[None]
        St = self.S[1:, 0].sum(0)[None] if self.S is not None else None
        post = op.posterior(Nt, st, St, nthreads=1)
        self.draw_epoch += 1
        par = op.sample(post, self.seed, self.draw_epoch, np.arange(1), nthreads=1)
        if self.leader_mode:
            for k in sorted(par):
                self.comm.broadcast(par[k])
        self._outlier_params = {k: np.array(v[0]) for k, v in par.items()}

    def init_first_clusters(self, init_clusters):
        """init_model_from_data labels (dp-parallel-sampling.jl:49-50) + init_first_clusters! (:62-78)."""
        out = 1 if self.outlier_weight > 0 else 0
        self._alloc(int(init_clusters) + out)
        self.wk.init_labels(int(init_clusters), self._next_epoch())
        if out:                                              # rand(1:initial_clusters) .+ 1 (dp-parallel-sampling.jl:49)
            lab, _ = self.wk.get_labels()
            self.wk.set_labels(lab + 1, None)
        self.wk.reset_sublabels(None, self._next_epoch())   # split_first_cluster_worker!
        self.wk.set_num_clusters(self.K)
        self.update_suff_stats_posterior()
        if out:
            self._create_outlier_cluster()
        if self.smart_splits:                                # dp-parallel-sampling.jl:70-75
            for k in range(self.K):
                self.smart_cluster_init(k)
            self.update_suff_stats_posterior()
        self.sample_clusters()

    def start_from_labels(self, labels, sub_labels, K):
        """Resume / benchmark entry: adopt given (local shard) labels instead of random ones."""
        self._alloc(int(K))
        self.wk.set_labels(labels, sub_labels)
        self.wk.set_num_clusters(self.K)
        self.update_suff_stats_posterior()
        self.sample_clusters()

    def log_posterior(self):
        """calculate_posterior (dp-parallel-sampling.jl:458-470)."""
        K = self.K
        L = self.prior.log_marginal(self.post, self.N.reshape(3 * K)).reshape(K, 3)[:, 0]
        Nc = self.N[:, 0]
        lp = gammaln(self.alpha) - gammaln(self.n_total + self.alpha)
        nz = Nc > 0
        return float(lp + np.sum(L[nz] + np.log(self.alpha) + gammaln(Nc[nz])))

    def run_model(self, iterations, first_iter=1, verbose=False, gt=None, on_iteration=None):
        """run_model (dp-parallel-sampling.jl:336-404): returns iter_count, nmi_history, likelihood_history, cluster_count_history."""
        iter_count, nmi_hist, lik_hist, k_hist = [], [], [], []
        self.vi_history = []
        if gt is not None:
            # ground truth of this shard goes to the GPU once; per iteration only a K x n_gt table comes back
            ids, inv = np.unique(np.asarray(gt), return_inverse=True)
            lo = getattr(self.wk, "first_index", 0)
            self.wk.set_ground_truth_range(inv[lo:lo + self.wk.n], len(ids))
        for i in range(first_iter, iterations + 1):
            final = i >= iterations - self.argmax_sample_stop
            no_more_splits = (i >= iterations - self.split_stop) or (self.K >= self.max_clusters)
            t0 = time.perf_counter()
            self.group_step(no_more_splits, final)
            dt = time.perf_counter() - t0
            iter_count.append(dt)
            k_hist.append(self.K)
            if gt is not None:
                nmi, vi = nmi_vi_from_contingency(self.comm.reduce_counts(self.wk.contingency(self.K)))
                nmi_hist.append(nmi); self.vi_history.append(vi)
            else:
                nmi_hist.append("no gt"); self.vi_history.append("no gt")
            if verbose:
                lik_hist.append(self.log_posterior())
                if self.comm.rank == 0:
                    print(f"Iteration: {i} || Clusters count: {self.K} || Log posterior: {lik_hist[-1]} || NMI score: {nmi_hist[-1]}"
                          f" || Iter Time:{dt} || Total time:{sum(iter_count)}")
            else:
                lik_hist.append(1)
            if on_iteration is not None:
                on_iteration(i, self)
        return iter_count, nmi_hist, lik_hist, k_hist

