"""ctypes front-end of the CPU oracle + numpy restatement of the host-side maths.

TEST INFRASTRUCTURE ONLY (see oracle/dpmm_oracle.c header).  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product package.

Reference files restated here (paths relative to the reference checkout):
  src/priors/niw.jl:20-31,53-62      calc_posterior / log_marginal_likelihood (NIW)
  src/priors/multinomial_prior.jl:16-21,34-39
  src/utils.jl:66-72                 log_multivariate_gamma (Float32 accumulator quirk)
  src/local_clusters_actions.jl:318-343  split Hastings ratio
  src/shared_actions.jl:21-38        merge Hastings ratio
  src/dp-parallel-sampling.jl:458-470 calculate_posterior
The worker-side functions live in dpmm_oracle.c; `sweep_numpy_*` below is the BLAS-backed
restatement of the same worker path with the reference's own structure (per-cluster GEMM +
column dot, materialised n x K table, gathered sub-cluster views, three Float64 statistic
passes per cluster) and is what bench.py times as `cpu_baseline` (kind "port").
"""
import ctypes
import os
import subprocess

import numpy as np
from scipy.special import gammaln

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dpmm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_exp_det.restype = ctypes.c_float
        _LIB.orc_exp_det.argtypes = [ctypes.c_float]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------- RNG
def philox(seed, idx, epoch, stream):
    out = (ctypes.c_uint32 * 4)()
    lib().orc_philox(ctypes.c_uint64(seed), ctypes.c_uint64(idx), ctypes.c_uint32(epoch), ctypes.c_uint32(stream), out)
    return [int(v) for v in out]


def uniforms(seed, epoch, stream, first_idx, n):
    u0 = np.empty(n, np.float32)
    u1 = np.empty(n, np.float32)
    lib().orc_uniforms(ctypes.c_uint64(seed), ctypes.c_uint32(epoch), ctypes.c_uint32(stream),
                       ctypes.c_int64(first_idx), ctypes.c_int64(n), _p(u0, c_f32p), _p(u1, c_f32p))
    return u0, u1


def exp_det(x):
    return np.array([lib().orc_exp_det(float(v)) for v in np.atleast_1d(x)], np.float32)


# --------------------------------------------------------------------------- worker path (C)
def _xinfo(X):
    """X: (n, ld) C-contiguous float32 whose row i is point i (== Julia D x n column-major)."""
    assert X.dtype == np.float32 and X.flags.c_contiguous and X.ndim == 2
    return X.shape[0], X.shape[1]


def niw_loglik_ref(X, D, mu, invS, logdet):
    n, ld = _xinfo(X)
    out = np.empty(n, np.float32)
    mu = _f32(mu); invS = _f32(invS)
    lib().orc_niw_loglik_ref(_p(X, c_f32p), D, ctypes.c_int64(n), ctypes.c_int64(ld), _p(mu, c_f32p),
                             _p(invS, c_f32p), ctypes.c_float(logdet), _p(out, c_f32p))
    return out


def niw_loglik_f64(X, D, mu, invS, logdet):
    n, ld = _xinfo(X)
    out = np.empty(n, np.float64)
    mu = _f32(mu); invS = _f32(invS)
    lib().orc_niw_loglik_f64(_p(X, c_f32p), D, ctypes.c_int64(n), ctypes.c_int64(ld), _p(mu, c_f32p),
                             _p(invS, c_f32p), ctypes.c_float(logdet), _p(out, c_f64p))
    return out


def mult_loglik_ref(X, D, logp):
    n, ld = _xinfo(X)
    out = np.empty(n, np.float32)
    logp = _f32(logp)
    lib().orc_mult_loglik_ref(_p(X, c_f32p), D, ctypes.c_int64(n), ctypes.c_int64(ld), _p(logp, c_f32p), _p(out, c_f32p))
    return out


def sample_log_cat(parr, u):
    """parr: (K, n) float32; u: (n,) float32 -> 1-based labels (int64)."""
    parr = _f32(parr); u = _f32(u)
    K, n = parr.shape
    lab = np.empty(n, np.int64)
    lib().orc_sample_log_cat(_p(parr, c_f32p), ctypes.c_int64(n), K, _p(u, c_f32p), _p(lab, c_i64p))
    return lab


def argmax_rows(parr):
    parr = _f32(parr)
    K, n = parr.shape
    lab = np.empty(n, np.int64)
    lib().orc_argmax_rows(_p(parr, c_f32p), ctypes.c_int64(n), K, _p(lab, c_i64p))
    return lab


def sweep_niw(X, D, mu, invS, logdet, logw, loglr, seed, epoch, first_idx=0, final=False, want_parr=False):
    n, ld = _xinfo(X)
    K = len(logw)
    mu = _f32(mu); invS = _f32(invS); logdet = _f32(logdet); logw = _f32(logw); loglr = _f32(loglr)
    assert mu.shape == (3 * K, D) and invS.shape == (3 * K, D * D) and logdet.shape == (3 * K,) and loglr.shape == (K, 2)
    lab = np.empty(n, np.int64); sub = np.empty(n, np.int64)
    parr = np.empty((K, n), np.float32) if want_parr else None
    lib().orc_sweep_niw(_p(X, c_f32p), D, ctypes.c_int64(n), ctypes.c_int64(ld), K, _p(mu, c_f32p), _p(invS, c_f32p),
                        _p(logdet, c_f32p), _p(logw, c_f32p), _p(loglr, c_f32p), ctypes.c_uint64(seed),
                        ctypes.c_uint32(epoch), ctypes.c_int64(first_idx), int(final), _p(lab, c_i64p),
                        _p(sub, c_i64p), _p(parr, c_f32p))
    return (lab, sub, parr) if want_parr else (lab, sub)


def sweep_mult(X, D, logp, logw, loglr, seed, epoch, first_idx=0, final=False, want_parr=False):
    n, ld = _xinfo(X)
    K = len(logw)
    logp = _f32(logp); logw = _f32(logw); loglr = _f32(loglr)
    assert logp.shape == (3 * K, D) and loglr.shape == (K, 2)
    lab = np.empty(n, np.int64); sub = np.empty(n, np.int64)
    parr = np.empty((K, n), np.float32) if want_parr else None
    lib().orc_sweep_mult(_p(X, c_f32p), D, ctypes.c_int64(n), ctypes.c_int64(ld), K, _p(logp, c_f32p),
                         _p(logw, c_f32p), _p(loglr, c_f32p), ctypes.c_uint64(seed), ctypes.c_uint32(epoch),
                         ctypes.c_int64(first_idx), int(final), _p(lab, c_i64p), _p(sub, c_i64p), _p(parr, c_f32p))
    return (lab, sub, parr) if want_parr else (lab, sub)


def suffstats_niw(X, D, labels, sub, K):
    n, ld = _xinfo(X)
    labels = np.ascontiguousarray(labels, np.int64); sub = np.ascontiguousarray(sub, np.int64)
    N = np.empty(3 * K); s = np.empty((3 * K, D)); S = np.empty((3 * K, D, D))
    lib().orc_suffstats_niw(_p(X, c_f32p), D, ctypes.c_int64(n), ctypes.c_int64(ld), _p(labels, c_i64p),
                            _p(sub, c_i64p), K, _p(N, c_f64p), _p(s, c_f64p), _p(S, c_f64p))
    return N.reshape(K, 3), s.reshape(K, 3, D), S.reshape(K, 3, D, D)


def suffstats_mult(X, D, labels, sub, K):
    n, ld = _xinfo(X)
    labels = np.ascontiguousarray(labels, np.int64); sub = np.ascontiguousarray(sub, np.int64)
    N = np.empty(3 * K, np.float32); s = np.empty((3 * K, D), np.float32)
    lib().orc_suffstats_mult(_p(X, c_f32p), D, ctypes.c_int64(n), ctypes.c_int64(ld), _p(labels, c_i64p),
                             _p(sub, c_i64p), K, _p(N, c_f32p), _p(s, c_f32p))
    return N.reshape(K, 3), s.reshape(K, 3, D)


def init_labels(n, init_clusters, seed, epoch, first_idx=0):
    lab = np.empty(n, np.int64); sub = np.empty(n, np.int64)
    lib().orc_init_labels(_p(lab, c_i64p), _p(sub, c_i64p), ctypes.c_int64(n), init_clusters, ctypes.c_uint64(seed),
                          ctypes.c_uint32(epoch), ctypes.c_int64(first_idx))
    return lab, sub


def split_relabel(labels, sub, idx, new_idx, seed, epoch, first_idx=0):
    idx = np.ascontiguousarray(idx, np.int64); new_idx = np.ascontiguousarray(new_idx, np.int64)
    lib().orc_split_relabel(_p(labels, c_i64p), _p(sub, c_i64p), ctypes.c_int64(len(labels)), _p(idx, c_i64p),
                            _p(new_idx, c_i64p), len(idx), ctypes.c_uint64(seed), ctypes.c_uint32(epoch),
                            ctypes.c_int64(first_idx))


def merge_relabel(labels, sub, idx, new_idx):
    idx = np.ascontiguousarray(idx, np.int64); new_idx = np.ascontiguousarray(new_idx, np.int64)
    lib().orc_merge_relabel(_p(labels, c_i64p), _p(sub, c_i64p), ctypes.c_int64(len(labels)), _p(idx, c_i64p),
                            _p(new_idx, c_i64p), len(idx))


def remove_empty(labels, pts_count):
    pc = np.ascontiguousarray(pts_count, np.int64)
    lib().orc_remove_empty(_p(labels, c_i64p), ctypes.c_int64(len(labels)), _p(pc, c_i64p), len(pc))


def reset_sub(labels, sub, idx, seed, epoch, first_idx=0):
    if idx is None:
        lib().orc_reset_sub(_p(labels, c_i64p), _p(sub, c_i64p), ctypes.c_int64(len(labels)), None, 0,
                            ctypes.c_uint64(seed), ctypes.c_uint32(epoch), ctypes.c_int64(first_idx))
        return
    idx = np.ascontiguousarray(idx, np.int64)
    lib().orc_reset_sub(_p(labels, c_i64p), _p(sub, c_i64p), ctypes.c_int64(len(labels)), _p(idx, c_i64p), len(idx),
                        ctypes.c_uint64(seed), ctypes.c_uint32(epoch), ctypes.c_int64(first_idx))


# --------------------------------------------------------------------------- host maths (numpy)
def niw_calc_posterior(kappa, m, nu, psi, N, points_sum, S):
    """priors/niw.jl:20-31.  kappa/nu are Float32 in the reference; m, psi Float64."""
    if N == 0:
        return kappa, np.array(m, float), nu, np.array(psi, float)
    k = np.float32(np.float32(kappa) + np.float32(N))
    v = np.float32(np.float32(nu) + np.float32(N))
    m = np.asarray(m, float); psi = np.asarray(psi, float)
    mp = (m * float(np.float32(kappa)) + points_sum) / float(k)
    pp = (float(np.float32(nu)) * psi + float(np.float32(kappa)) * np.outer(m, m) - float(k) * np.outer(mp, mp) + S) / float(v)
    pp = np.triu(pp) + np.triu(pp, 1).T  # Matrix(Symmetric(psi)) takes the upper triangle
    pp = (pp + pp.T) / 2
    return float(k), mp, float(v), pp


def log_multivariate_gamma(x, D, f32_quirk=True):
    """utils.jl:66-72.  The reference accumulates into a Float32-typed local."""
    if f32_quirk:
        res = np.float32(D * (D - 1) / 4 * np.log(np.pi))
        for d in range(1, D + 1):
            res = np.float32(res + gammaln(x + (1 - d) / 2))
        return float(res)
    return D * (D - 1) / 4 * np.log(np.pi) + sum(gammaln(x + (1 - d) / 2) for d in range(1, D + 1))


def niw_log_marginal(prior, post, N, D, f32_quirk=True):
    """priors/niw.jl:53-62.  prior/post = (kappa, m, nu, psi)."""
    k0, _, v0, p0 = prior
    k1, _, v1, p1 = post
    ld0 = np.linalg.slogdet(p0)[1]
    ld1 = np.linalg.slogdet(p1)[1]
    return (-N * D * 0.5 * np.log(np.pi) + log_multivariate_gamma(v1 / 2, D, f32_quirk)
            - log_multivariate_gamma(v0 / 2, D, f32_quirk) + (v0 / 2) * (D * np.log(v0) + ld0)
            - (v1 / 2) * (D * np.log(v1) + ld1) + (D / 2) * np.log(k0 / k1))


def niw_posterior_predictive(X, kappa, m, nu, psi):
    """priors/niw.jl:68-76: logpdf(MvTDist(nu-D+1, m, ((kappa+1)/(kappa (nu-D+1))) nu psi), x) via scipy (closed form
    independent of the build)."""
    from scipy.stats import multivariate_t
    D = len(m)
    df = nu - D + 1
    return multivariate_t(loc=m, shape=((kappa + 1) / (kappa * df)) * nu * np.asarray(psi), df=df).logpdf(np.asarray(X, float))


def mult_calc_posterior(alpha, N, points_sum):
    """priors/multinomial_prior.jl:16-21 (Float32 arithmetic)."""
    if N == 0:
        return np.asarray(alpha, np.float32)
    return (np.asarray(alpha, np.float32) + np.asarray(points_sum, np.float32)).astype(np.float32)


def mult_log_marginal(alpha, alpha_post):
    """priors/multinomial_prior.jl:34-39 (evaluated in Float64 here)."""
    a = np.asarray(alpha, float); b = np.asarray(alpha_post, float)
    return gammaln(a.sum()) - gammaln(b.sum()) + np.sum(gammaln(b) - gammaln(a))


def split_log_hr(alpha, N_l, L_l, N_r, L_r, N, L):
    """local_clusters_actions.jl:336-339."""
    return np.log(alpha) + gammaln(N_l) + L_l + gammaln(N_r) + L_r - (gammaln(N) + L)


def merge_log_hr(alpha, N_i, L_i, N_j, L_j, L):
    """shared_actions.jl:28-30."""
    N = N_i + N_j
    return (-np.log(alpha) + gammaln(alpha) - 2 * gammaln(0.5 * alpha) + gammaln(N) - gammaln(N + alpha)
            + gammaln(N_i + 0.5 * alpha) - gammaln(N_i) - gammaln(N_j) + gammaln(N_j + 0.5 * alpha) + L - L_i - L_j)


def log_posterior(alpha, N_total, Ns, Ls):
    """dp-parallel-sampling.jl:458-470."""
    lp = gammaln(alpha) - gammaln(N_total + alpha)
    for N, L in zip(Ns, Ls):
        if N == 0:
            continue
        lp += L + np.log(alpha) + gammaln(N)
    return lp


# --------------------------------------------------------------------------- BLAS-backed worker sweep (cpu_baseline)
def sweep_numpy_niw(X, D, mu, invS, logdet, logw, loglr, u0, u1, final=False):
    """One worker's sample_labels_worker! + sample_sub_clusters_worker! + create_suff_stats_dict_worker
    with the reference's structure: per-cluster GEMM invS*z (BLAS) and column dot
    (mv_gaussian.jl:21-25), n x K table, row-wise max-shift/exp/normalise + inverse-CDF scan
    (utils.jl:19-31), gathered views per cluster for sub-labels (local_clusters_actions.jl:77-78)
    and three Float64 statistic passes per cluster (:158-166, niw.jl:42-51).
    X is (n, D) float32 (row = point).  Returns labels, sub, (N, sum, S)."""
    n = X.shape[0]
    K = len(logw)
    Xt = np.ascontiguousarray(X.T)  # D x n like the reference
    parr = np.empty((n, K), np.float32)
    l2pi = np.float32(np.log(2 * np.pi))

    def loglik(pts, j):
        z = pts - mu[j][:, None]
        y = invS[j].reshape(D, D).T @ z
        r = np.einsum("ij,ij->j", z, y)
        return -((np.float32(D * D) * l2pi + logdet[j]) / np.float32(2)) - r / np.float32(2)

    for k in range(K):
        parr[:, k] = loglik(Xt, 3 * k) + logw[k]

    def draw(p, u):
        p = np.where(np.isnan(p), -np.inf, p)
        p = p - p.max(axis=1, keepdims=True)
        np.exp(p, out=p)
        p /= p.sum(axis=1, keepdims=True)
        cw = np.cumsum(p, axis=1)
        t = (u * cw[:, -1])[:, None]
        return np.minimum((cw < t).sum(axis=1), p.shape[1] - 1) + 1

    labels = (parr.argmax(axis=1) + 1) if final else draw(parr, u0)
    sub = np.empty(n, np.int64)
    for k in range(K):
        msk = labels == k + 1
        if not msk.any():
            continue
        pts = Xt[:, msk]
        p2 = np.stack([loglik(pts, 3 * k + 1) + loglr[k, 0], loglik(pts, 3 * k + 2) + loglr[k, 1]], axis=1)
        sub[msk] = draw(p2, u1[msk])
    Ns = np.zeros((K, 3)); sums = np.zeros((K, 3, D)); Ss = np.zeros((K, 3, D, D))
    for k in range(K):
        msk = labels == k + 1
        pts = Xt[:, msk]; sl = sub[msk]
        for w, sel in enumerate((slice(None), sl == 1, sl == 2)):
            p = pts[:, sel].astype(np.float64)
            Ns[k, w] = p.shape[1]
            if p.shape[1]:
                sums[k, w] = p.sum(axis=1)
                S = p @ p.T
                Ss[k, w] = 0.5 * (S + S.T)
    return labels, sub, (Ns, sums, Ss)
