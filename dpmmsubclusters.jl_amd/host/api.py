"""Entry points mirroring the reference's user API (src/DPMMSubClusters.jl:36 exports):

    fit(all_data, [hyper_params], alpha; iters, init_clusters, seed, verbose, save_model, burnout,
        gt, max_clusters, outlier_weight, outlier_params, smart_splits)     dp-parallel-sampling.jl:215-293
    dp_parallel(all_data, hyper_params, alpha, iters, init_clusters, seed, verbose, save_model,
        burnout, gt, max_clusters, outlier_weight, outlier_params, smart_splits)          :121-157
    generate_gaussian_data / generate_mnmm_data   (data_generators.jl:19-72; build-owned recipes)

Argument meaning, defaults, coercions (Float32 data / Int64 iters, :279-293), the default NIW prior
(kappa=1, m=0, nu=D+3, psi=I, :272-274) and the 9-tuple / 5-tuple results follow the reference.
`all_data` is Dimensions x Samples (D x N) as in the reference.  Checkpoints (`save_model`), the advanced
parameter-file mode `dp_parallel(model_params::String)` and `run_model_from_checkpoint` live in host/checkpoint.py.
Smart splits (`smart_splits=True`, Gaussian prior) are driven by DPMMSampler.smart_cluster_init; the outlier component
(`outlier_weight`, `outlier_params`) is cluster 1 of the model with a constant weight, never split, merged or re-drawn.

Distributed: when torch.distributed is initialised (one process per GPU) every rank calls `fit`
with the SAME full arguments; each rank keeps the contiguous column range
[rank*N/W, (rank+1)*N/W) of the data on its GPU (the DArray layout of `distribute`,
dp-parallel-sampling.jl:42-50) and the per-sweep exchange is one all-reduce of the packed
sufficient statistics (host/comm.py).
"""
import numpy as np

from .. import binding
from . import priors as _priors
from . import checkpoint as _ckpt
from .priors import multinomial_hyper, niw_hyperparams
from .sampler import DPMMSampler, LocalComm


class dp_parallel_sampling:
    """Result handle (the reference's `dp_parallel_sampling` struct, src/ds.jl:75-78, reduced to what
    callers use): hyper-parameters, alpha, the sampler with cluster state, and labels."""

    def __init__(self, sampler, labels, sub_labels):
        self.sampler = sampler
        self.model_hyperparams = dict(distribution_hyper_params=sampler.prior, alpha=sampler.alpha, total_dim=sampler.n_total)
        self.labels = labels
        self.labels_subcluster = sub_labels

    @property
    def num_clusters(self):
        return self.sampler.K


def _shard(N, comm):
    lo = (N * comm.rank) // comm.world
    hi = (N * (comm.rank + 1)) // comm.world
    return lo, hi


def _make_sampler(all_data, hyper, alpha, seed, burnout, max_clusters, comm, device, nthreads=None, worker_factory=None,
                  rows=None, **sampler_kw):
    """`all_data`: Dimensions x Samples (basic mode), or `rows`: Samples x Dimensions as stored in a .npy file (advanced
    mode; cleaned and converted on the GPU by dpmm_upload_points_npy)."""
    if rows is not None:
        N, D = rows.shape
    else:
        X = np.asarray(all_data)
        if X.ndim != 2:
            raise ValueError("all_data must be Dimensions x Samples")
        D, N = X.shape
    if hyper.dim != D:
        raise ValueError(f"prior dimension {hyper.dim} != data dimension {D}")
    lo, hi = _shard(N, comm)
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1)[0])
        seed = comm.broadcast_int(seed) if hasattr(comm, "broadcast_int") else seed
    kw = dict(timing=False) if worker_factory is None else {}      # the product path records no timing events (~5 us each, four per step)
    wk = (worker_factory or binding.Worker)(hyper.kind, D, hi - lo, first_index=lo, device=device, seed=int(seed), **kw)
    if rows is not None:
        if hasattr(wk, "upload_points_npy"):
            wk.upload_points_npy(rows[lo:hi])
        else:
            wk.upload_points(np.nan_to_num(np.asarray(rows[lo:hi], dtype=np.float32), nan=0.0, posinf=np.inf, neginf=-np.inf))
    else:
        wk.upload_points(np.ascontiguousarray(X[:, lo:hi].T, dtype=np.float32))  # (n_local, D): row = point
    return DPMMSampler(wk, hyper, alpha, N, int(seed), burnout=burnout, max_clusters=max_clusters, comm=comm, nthreads=nthreads,
                       **sampler_kw)


def _check_next_rows(outlier_weight, outlier_params, smart_splits, hyper=None):
    if outlier_weight and outlier_weight > 0:
        if not isinstance(outlier_params, _priors.distribution_hyper_params):
            raise TypeError("outlier_weight > 0 needs outlier_params (a distribution_hyper_params of the same family)")
        if hyper is not None and (outlier_params.kind != hyper.kind or outlier_params.dim != hyper.dim):
            raise ValueError("outlier_params must be of the same family and dimension as the cluster prior")
        if not (outlier_weight < 1):
            raise ValueError("outlier_weight must be in (0, 1)")
    if smart_splits and hyper is not None and hyper.kind != _priors.PRIOR_NIW:
        raise ValueError("smart_splits is available for the Gaussian (niw_hyperparams) prior only, as in the reference")


def _comm_device(comm, device):
    if comm is None:
        from .comm import default_comm
        comm = default_comm()
    if device is None:
        device = getattr(comm, "device", 0)
    return comm, device


def dp_parallel(all_data, local_hyper_params=None, alpha_param=None, iters=100, init_clusters=1, seed=None, verbose=True,
                save_model=False, burnout=15, gt=None, max_clusters=np.inf, outlier_weight=0, outlier_params=None,
                smart_splits=False, comm=None, device=None, nthreads=None, worker_factory=None, save_path="./",
                save_file_prefix="checkpoint_", model_save_interval=1000):
    """dp_parallel(all_data, hyper_params, alpha, ...) -- basic mode (dp-parallel-sampling.jl:121-157), or
    dp_parallel(model_params::String; verbose, gt) -- advanced mode driven by a parameter file (:178-196).
    Returns (dp_model, iter_count, nmi_score_history, likelihood_history, cluster_count_history).
    `save_model=True` writes a checkpoint every `model_save_interval` iterations (global_params.jl:36-41 defaults)."""
    if isinstance(all_data, (str, bytes)) or hasattr(all_data, "__fspath__"):
        return _dp_parallel_from_params(str(all_data), verbose=verbose, gt=gt, comm=comm, device=device, nthreads=nthreads,
                                        worker_factory=worker_factory)
    if not isinstance(local_hyper_params, _priors.distribution_hyper_params):
        raise TypeError("local_hyper_params must be a distribution_hyper_params (niw_hyperparams / multinomial_hyper)")
    _check_next_rows(outlier_weight, outlier_params, smart_splits, local_hyper_params)
    comm, device = _comm_device(comm, device)
    s = _make_sampler(all_data, local_hyper_params, np.float32(alpha_param), seed, int(burnout), max_clusters, comm, device,
                      nthreads, worker_factory)
    s.smart_splits = bool(smart_splits)
    if outlier_weight and outlier_weight > 0:
        s.outlier_weight, s.outlier_prior = float(outlier_weight), outlier_params
    s.init_first_clusters(int(init_clusters))
    hook = _ckpt.SaveHook(save_path, save_file_prefix, model_save_interval, "none", 0.0, verbose) if save_model else None
    iter_count, nmi, lik, kh = s.run_model(int(iters), 1, verbose=verbose, gt=gt, on_iteration=hook)
    labels, sub = comm.gather_labels(s.wk)
    model = dp_parallel_sampling(s, labels, sub)
    model.checkpoints = hook.files if hook else []
    return model, iter_count, nmi, lik, kh


def _sampler_from_params(P, comm, device, nthreads, worker_factory):
    use_outlier = P["outlier_hyper_params"] is not None and P["outlier_mod"] and P["outlier_mod"] > 0
    _check_next_rows(P["outlier_mod"] if use_outlier else 0, P["outlier_hyper_params"], P["smart_splits"], P["hyper_params"])
    rows = _ckpt.load_data(P["data_path"], P["data_prefix"], swapDimension=False, mmap=True)     # Samples x Dimensions
    s = _make_sampler(None, P["hyper_params"], np.float32(P["alpha"]), P["random_seed"], int(P["burnout_period"]),
                      P["max_clusters"], comm, device, nthreads, worker_factory, rows=rows,
                      argmax_sample_stop=int(P["argmax_sample_stop"]), split_stop=int(P["split_stop"]))
    s.hard_clustering = bool(P["hard_clustering"])
    s.smart_splits = bool(P["smart_splits"])
    s.max_split_iter = int(P["max_split_iter"])
    if use_outlier:
        s.outlier_weight, s.outlier_prior = float(P["outlier_mod"]), P["outlier_hyper_params"]
    return s


def _run_with_params(s, P, first_iter, prev_time, model_params, verbose, gt):
    hook = None
    if P["enable_saving"]:
        hook = _ckpt.SaveHook(P["save_path"], P["save_file_prefix"], P["model_save_interval"], model_params, prev_time, verbose)
    iter_count, nmi, lik, kh = s.run_model(int(P["iterations"]), first_iter, verbose=verbose, gt=gt, on_iteration=hook)
    labels, sub = s.comm.gather_labels(s.wk)
    model = dp_parallel_sampling(s, labels, sub)
    model.checkpoints = hook.files if hook else []
    return model, iter_count, nmi, lik, kh


def _dp_parallel_from_params(model_params, verbose=True, gt=None, comm=None, device=None, nthreads=None, worker_factory=None):
    P = _ckpt.read_params(model_params)
    comm, device = _comm_device(comm, device)
    s = _sampler_from_params(P, comm, device, nthreads, worker_factory)
    s.init_first_clusters(int(P["initial_clusters"]))
    return _run_with_params(s, P, 1, 0.0, model_params, verbose, gt)


def run_model_from_checkpoint(filename, verbose=True, gt=None, comm=None, device=None, nthreads=None, worker_factory=None):
    """run_model_from_checkpoint(filename) (dp-parallel-sampling.jl:428-447): load the point-less group, re-read the
    parameter file it names, load the data from the same path, restore labels / cluster state and continue at iter+1.
    Every rank calls it with the same file.  Returns the 5-tuple of dp_parallel."""
    ck = _ckpt.load_checkpoint(filename)
    gp = str(ck["global_params"])
    if gp == "none" or not gp:
        raise ValueError("this checkpoint was written in basic mode (fit / dp_parallel with arrays): it has no parameter "
                         "file to reload the data from; use resume_from_checkpoint(filename, all_data, ...) instead")
    P = _ckpt.read_params(gp)
    comm, device = _comm_device(comm, device)
    if P["random_seed"] is None:
        P["random_seed"] = int(ck["seed"])
    s = _sampler_from_params(P, comm, device, nthreads, worker_factory)
    _ckpt.restore_sampler(s, ck)
    return _run_with_params(s, P, int(ck["iter"]) + 1, float(ck["total_time"]), gp, verbose, gt)


def resume_from_checkpoint(filename, all_data, iters, verbose=True, gt=None, burnout=None, max_clusters=np.inf, comm=None,
                           device=None, nthreads=None, worker_factory=None, save_model=False, save_path="./",
                           save_file_prefix="checkpoint_", model_save_interval=1000):
    """Basic-mode counterpart of run_model_from_checkpoint: the caller supplies the data array again (D x N) and the
    total number of iterations; the chain continues at iter+1 exactly where the checkpoint left it."""
    ck = _ckpt.load_checkpoint(filename)
    comm, device = _comm_device(comm, device)
    hyper = _ckpt._prior_from_dict(ck)
    s = _make_sampler(all_data, hyper, np.float32(ck["alpha"]), int(ck["seed"]), int(ck["burnout"] if burnout is None else burnout),
                      max_clusters, comm, device, nthreads, worker_factory)
    _ckpt.restore_sampler(s, ck)
    hook = _ckpt.SaveHook(save_path, save_file_prefix, model_save_interval, "none", float(ck["total_time"]), verbose) if save_model else None
    iter_count, nmi, lik, kh = s.run_model(int(iters), int(ck["iter"]) + 1, verbose=verbose, gt=gt, on_iteration=hook)
    labels, sub = comm.gather_labels(s.wk)
    model = dp_parallel_sampling(s, labels, sub)
    model.checkpoints = hook.files if hook else []
    return model, iter_count, nmi, lik, kh


def fit(all_data, *args, iters=100, init_clusters=1, seed=None, verbose=True, save_model=False, burnout=20, gt=None,
        max_clusters=np.inf, outlier_weight=0, outlier_params=None, smart_splits=False, **kw):
    """fit(all_data, alpha; ...) or fit(all_data, hyper_params, alpha; ...).

    Returns the reference's 9-tuple: (labels, clusters, weights, iter_count, nmi_score_history,
    likelihood_history, cluster_count_history, sub_labels, dp_model)."""
    if len(args) == 1:
        D = np.asarray(all_data).shape[0]
        hyper = niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))   # dp-parallel-sampling.jl:272-274
        alpha = args[0]
    elif len(args) == 2:
        hyper, alpha = args
    else:
        raise TypeError("fit(all_data, alpha; ...) or fit(all_data, hyper_params, alpha; ...)")
    dp_model, iter_count, nmi, lik, kh = dp_parallel(all_data, hyper, alpha, int(iters), int(init_clusters), seed, verbose,
                                                     save_model, burnout, gt, max_clusters, outlier_weight, outlier_params,
                                                     smart_splits, **kw)
    s = dp_model.sampler
    clusters = s.prior.distributions(s.params, [3 * k for k in range(s.K)])
    return (dp_model.labels, clusters, s.weights, iter_count, nmi, lik, kh, dp_model.labels_subcluster, dp_model)


def predict(dp_model, data, device=None, worker_factory=None):
    """predict(dp_model, data) -- src/dp-parallel-sampling.jl:532-537 with predict_points
    (src/local_clusters_actions.jl:23-40): weights = (points_count + alpha) / sum; per cluster the posterior predictive
    log-density (GPU), labels = row-wise argmax, probabilities = normalised exponentials (NaN -> -Inf).
    `data` is Dimensions x Samples.  Returns (labels (n,) Int64 1-based, probs (n, K) Float32)."""
    s = dp_model.sampler
    post = s.post
    X = np.ascontiguousarray(np.asarray(data, dtype=np.float32).T)
    n, D = X.shape
    if D != s.prior.dim:
        raise ValueError("data dimension does not match the model")
    w = s.points_count.astype(np.float64) + s.alpha
    w = (w / w.sum()).astype(np.float32)
    dev = getattr(s.wk, "device", 0) if device is None else device
    wk = (worker_factory or binding.Worker)(s.prior.kind, D, n, first_index=0, device=dev, seed=0)
    try:
        wk.upload_points(X)
        if getattr(wk, "supports_predict_points", False):     # argmax + normalisation on the device as well
            return s.prior.predictive_table(wk, post, [3 * k for k in range(s.K)], w, points=True)
        parr = s.prior.predictive_table(wk, post, [3 * k for k in range(s.K)], w).T.astype(np.float32)   # (n, K)
    finally:
        wk.close()
    with np.errstate(invalid="ignore"):
        has_nan = np.isnan(parr).any(1)
        lbls = np.where(has_nan, np.isnan(parr).argmax(1), parr.argmax(1)) + 1   # Julia's argmax returns the first NaN
        parr = np.where(np.isnan(parr), -np.inf, parr)
        parr = parr - parr.max(1, keepdims=True)
        np.exp(parr, out=parr)
        parr /= parr.sum(1, keepdims=True)
    return lbls.astype(np.int64), parr


def get_labels_histogram(labels):
    """utils.jl:39-48: sorted [(label, count)]"""
    v, c = np.unique(np.asarray(labels), return_counts=True)
    return list(zip(v.tolist(), c.tolist()))


# ----------------------------------------------------------------------------- synthetic inputs
def _mixture_spec(N, D, K, MixtureVar, seed):
    """Component sizes, means and covariance factors of the synthetic Gaussian mixture (the recipe of data_generators.jl:19-42:
    weights ~ Dir(1_K); sizes ~ Multinomial(N, weights); mean_k ~ N(0, MixtureVar I); cov_k ~ InvWishart(D+2, I))."""
    rng = np.random.default_rng([int(seed), 0])
    sizes = rng.multinomial(N, rng.dirichlet(np.ones(K)))
    means = rng.normal(size=(K, D)) * np.sqrt(MixtureVar)
    chol = np.empty((K, D, D))
    for k in range(K):
        G = rng.normal(size=(D + 2, D))                       # inverse of a Wishart(D+2, I) draw
        chol[k] = np.linalg.cholesky(np.linalg.inv(G.T @ G))
    return sizes, means, chol


def gaussian_mixture_shard(N, D, K, MixtureVar, seed, lo, hi, chunk=100000):
    """Columns [lo, hi) of the N-point synthetic mixture of `_mixture_spec`, points of a component contiguous (as the
    reference's generator lays them out), generated chunk-wise with per-chunk seeds so that every rank of a multi-GPU run
    builds only its own column range (the dataset does not depend on the number of ranks).
    Returns (X (hi-lo, D) float32 row = point, labels (hi-lo,) int64 1-based)."""
    sizes, means, chol = _mixture_spec(N, D, K, MixtureVar, seed)
    edges = np.concatenate([[0], np.cumsum(sizes)])
    X = np.empty((hi - lo, D), np.float32)
    lab = np.empty(hi - lo, np.int64)
    c0, c1 = lo // chunk, (hi - 1) // chunk if hi > lo else -1
    for c in range(c0, c1 + 1):
        a, b = c * chunk, min((c + 1) * chunk, N)
        z = np.random.default_rng([int(seed), 1 + c]).standard_normal((b - a, D), dtype=np.float32)
        comp = np.searchsorted(edges, np.arange(a, b), side="right") - 1
        out = np.empty((b - a, D), np.float32)
        for k in np.unique(comp):
            m = comp == k
            out[m] = (means[k] + z[m] @ chol[k].T.astype(np.float32)).astype(np.float32)
        s, e = max(a, lo), min(b, hi)
        X[s - lo:e - lo] = out[s - a:e - a]
        lab[s - lo:e - lo] = comp[s - a:e - a] + 1
    return X, lab


def generate_gaussian_data(N, D, K, MixtureVar, seed=None):
    """generate_gaussian_data(N, D, K, MixtureVar) (data_generators.jl:19-42), same return shape:
    (x D x N Float32, labels (N,) Float32 1-based, means D x K, covariances D x D x K).  The whole range of
    `gaussian_mixture_shard`, so single- and multi-rank runs see the same points."""
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1)[0])
    X, lab = gaussian_mixture_shard(int(N), int(D), int(K), MixtureVar, seed, 0, int(N))
    _, means, chol = _mixture_spec(int(N), int(D), int(K), MixtureVar, seed)
    covs = np.einsum("kab,kcb->ack", chol, chol)
    return np.ascontiguousarray(X.T), lab.astype(np.float32), means.T.astype(np.float32), covs.astype(np.float32)


def generate_mnmm_data(N, D, K, trials, seed=None):
    """Recipe of data_generators.jl:59-72. Returns (x D x N f32 counts, labels, clusters D x K)."""
    rng = np.random.default_rng(seed)
    clusters = np.zeros((D, K))
    labels = rng.integers(1, K + 1, N)
    for i in range(K):
        alphas = rng.integers(1, 21, D).astype(float)
        alphas[i % D] = rng.integers(30, 101)
        clusters[:, i] = rng.dirichlet(alphas)
    x = np.empty((D, N), np.float32)
    for i in range(K):
        m = labels == i + 1
        x[:, m] = rng.multinomial(trials, clusters[:, i], size=int(m.sum())).T
    return x, labels, clusters
