"""TEST-ONLY stand-in for the GPU `Worker` (dpmmsubclusters.jl_amd/binding.py) built on the CPU oracle.
It exists so that the host sampler and the multi-rank exchange can be exercised on machines
without a GPU (world_size-2 gloo tests).  It is never importable from the product package."""
import numpy as np

from oracle import oracle as orc

PRIOR_NIW, PRIOR_MULT = 0, 1


class FakeWorker:
    def __init__(self, prior, D, n_local, first_index=0, device=0, seed=0):
        self.prior, self.D, self.n, self.first_index, self.seed = prior, int(D), int(n_local), int(first_index), int(seed)
        self.K = 0
        self.packed_stride = 1 + D + (D * (D + 1) // 2 if prior == PRIOR_NIW else 0)
        self.labels = np.ones(self.n, np.int64); self.sub = np.ones(self.n, np.int64)

    def close(self):
        pass

    def upload_points(self, X):
        self.X = np.ascontiguousarray(X, np.float32)

    def init_labels(self, init_clusters, epoch):
        self.labels, self.sub = orc.init_labels(self.n, init_clusters, self.seed, epoch, self.first_index)

    def init_labels_from(self, init_clusters, first_label, epoch):
        self.init_labels(init_clusters, epoch)
        self.labels += first_label - 1

    # ---- the engine's worker table (include/dpmm_host.h: dpmmh_worker) over this object
    def params_staging(self, slots):
        D = self.D
        old = getattr(self, "_staging", None)
        if old is not None and old["slots"] >= slots:
            return old
        cap = max(8, slots)
        niw = self.prior == PRIOR_NIW
        new = dict(slots=cap, mu=np.zeros((3 * cap, D), np.float32) if niw else None,
                   mat=np.zeros((3 * cap, D * (D + 1) // 2 if niw else D), np.float32), logdet=np.zeros(3 * cap, np.float32) if niw else None,
                   lr=np.zeros((cap, 2), np.float32), w=np.zeros(cap, np.float32), slot=np.zeros(cap, np.int32))
        if old is not None:
            for k in ("mu", "mat", "logdet", "lr", "w", "slot"):
                if new[k] is not None:
                    new[k][:len(old[k])] = old[k]
        self._staging = new
        return new

    def commit_params(self, K):
        st = self._staging
        rows = (3 * st["slot"][:K].astype(np.int64)[:, None] + np.arange(3)[None, :]).ravel()
        if self.prior == PRIOR_NIW:
            D = self.D
            R = np.zeros((len(rows), D, D), np.float32)            # staging rows hold the packed upper triangle (row r: columns r..D-1)
            iu = np.triu_indices(D)
            R[:, iu[0], iu[1]] = st["mat"][rows]
            self.set_params_niw_chol(st["mu"][rows], R, st["logdet"][rows], st["lr"][:K], st["w"][:K])
        else:
            self.set_params_mult(st["mat"][rows], st["lr"][:K], st["w"][:K])

    def step_stats(self, reset_epoch, global_counts=None):
        """dpmm_step_stats: occupancies (summed over the ranks by the caller) -> bad flags -> sub-label reset -> statistics."""
        counts = self.bin_counts() if global_counts is None else np.asarray(global_counts).reshape(self.K, 2)
        bad = ((counts[:, 0] == 0) | (counts[:, 1] == 0)).astype(np.uint8)
        if bad.any():
            self.reset_sublabels(np.flatnonzero(bad) + 1, reset_epoch)
        return self.suffstats_packed(None), bad

    def set_labels(self, labels=None, sub=None):
        if labels is not None:
            self.labels = np.array(labels, np.int64)
        if sub is not None:
            self.sub = np.array(sub, np.int64)

    def get_labels(self):
        return self.labels.copy(), self.sub.copy()

    def set_num_clusters(self, K):
        self.K = int(K)

    def set_params_niw_chol(self, mu, R, logdet, lr_weights, weights):
        K = len(weights)
        R = np.asarray(R, np.float32).reshape(3 * K, self.D, self.D).astype(np.float64)
        R = np.triu(R)
        self.invS = np.einsum("kji,kjl->kil", R, R).reshape(3 * K, -1).astype(np.float32)  # R'R
        self.mu = np.asarray(mu, np.float32); self.logdet = np.asarray(logdet, np.float32)
        self.logw = np.log(np.asarray(weights, np.float32)); self.loglr = np.log(np.asarray(lr_weights, np.float32))
        self.K = K

    def set_params_mult(self, logp, lr_weights, weights):
        self.logp = np.asarray(logp, np.float32)
        self.logw = np.log(np.asarray(weights, np.float32)); self.loglr = np.log(np.asarray(lr_weights, np.float32))
        self.K = len(weights)

    def sweep(self, epoch, final=False):
        if self.n == 0:
            return
        if self.prior == PRIOR_NIW:
            self.labels, self.sub = orc.sweep_niw(self.X, self.D, self.mu, self.invS, self.logdet, self.logw, self.loglr,
                                                  self.seed, epoch, self.first_index, final)
        else:
            self.labels, self.sub = orc.sweep_mult(self.X, self.D, self.logp, self.logw, self.loglr, self.seed, epoch,
                                                   self.first_index, final)

    def suffstats_packed(self, cluster_idx=None):
        K, D = self.K, self.D
        out = np.zeros((2 * K, self.packed_stride))
        if self.prior == PRIOR_NIW:
            N, s, S = orc.suffstats_niw(self.X, D, self.labels, self.sub, K)
        else:
            N, s = orc.suffstats_mult(self.X, D, self.labels, self.sub, K)
            N = N.astype(np.float64); s = s.astype(np.float64); S = None
        sel = range(K) if cluster_idx is None else [int(i) - 1 for i in cluster_idx]
        il = np.tril_indices(D)
        for k in sel:
            for w in (1, 2):
                row = out[2 * k + w - 1]
                row[0] = N[k, w]; row[1:1 + D] = s[k, w]
                if S is not None:
                    row[1 + D:] = S[k, w][il]
        return out

    def unpack(self, packed, K=None):
        K = self.K if K is None else K
        D = self.D
        N = np.zeros((K, 3)); s = np.zeros((K, 3, D))
        S = np.zeros((K, 3, D, D)) if self.prior == PRIOR_NIW else None
        il = np.tril_indices(D)
        for k in range(K):
            for w in (1, 2):
                row = packed[2 * k + w - 1]
                N[k, w] = row[0]; s[k, w] = row[1:1 + D]
                if S is not None:
                    M = np.zeros((D, D)); M[il] = row[1 + D:]
                    S[k, w] = M + np.tril(M, -1).T
            N[k, 0] = N[k, 1] + N[k, 2]; s[k, 0] = s[k, 1] + s[k, 2]
            if S is not None:
                S[k, 0] = S[k, 1] + S[k, 2]
        return (N, s, S) if S is not None else (N, s)

    def split(self, idx, new_idx, epoch):
        orc.split_relabel(self.labels, self.sub, idx, new_idx, self.seed, epoch, self.first_index)

    def merge(self, idx, new_idx):
        orc.merge_relabel(self.labels, self.sub, idx, new_idx)

    def remove_empty(self, pts_count):
        orc.remove_empty(self.labels, pts_count)

    def reset_sublabels(self, idx, epoch):
        orc.reset_sub(self.labels, self.sub, idx, self.seed, epoch, self.first_index)

    def set_ground_truth_range(self, gt, n_gt):
        self.gt = np.asarray(gt, np.int64); self.n_gt = int(n_gt)

    # smart splits (worker halves), straight numpy restatement of local_clusters_actions.jl:629-653
    def smart_project(self, cluster, v, mu):
        m = self.labels == cluster
        self._proj = np.full(self.n, np.nan)
        self._proj[m] = (self.X[m].astype(np.float64) - np.asarray(mu, np.float64)) @ np.asarray(v, np.float64)
        return self._proj[m].copy()

    def smart_kmeans_iter(self, cluster, m_lo, m_hi):
        t = self._proj[self.labels == cluster]
        side1 = np.abs(t - m_lo) < np.abs(t - m_hi)
        return np.array([t[side1].sum(), side1.sum(), t[~side1].sum(), (~side1).sum()], np.float64)

    def smart_assign(self, cluster, m_lo, m_hi):
        m = self.labels == cluster
        t = self._proj[m]
        self.sub[m] = np.where(np.abs(t - m_lo) < np.abs(t - m_hi), 1, 2)

    def bin_counts(self):
        out = np.zeros((self.K, 2), np.int64)
        np.add.at(out, (self.labels - 1, self.sub - 1), 1)
        return out

    def contingency(self, K=None):
        K = self.K if K is None else K
        out = np.zeros((K, self.n_gt), np.int64)
        np.add.at(out, (self.labels - 1, self.gt), 1)
        return out

    def sync(self):
        pass
