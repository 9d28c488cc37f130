"""Checkpoint / resume and the advanced (parameter-file) mode -- SURVEY.md section 8(f), rank 3.

Reference behaviour mirrored here (paths relative to the reference checkout):
  load_data(path; prefix, swapDimension)     src/utils.jl:5-14        .npy Samples x Dimensions, NaN -> 0, transposed
  init_model()                               src/dp-parallel-sampling.jl:11-33
  dp_parallel(model_params::String; ...)     src/dp-parallel-sampling.jl:178-196
  save_model / run_model's save hook         src/dp-parallel-sampling.jl:398-403, 450-455
  run_model_from_checkpoint(filename)        src/dp-parallel-sampling.jl:428-447
  pts_less_group / create_pts_less_group     src/ds.jl:60-66, 85-92
  the parameter file                         src/global_params.jl, docs/src/usage.md:44-72

Differences that cannot be avoided without a Julia runtime, stated once:
  * the reference's parameter file is Julia source that it `include`s; here it is Python source executed with the same
    variable names in scope (`α` may also be spelled `alpha`; `DPMMSubClusters.niw_hyperparams`, `Inf`, `nothing`,
    `zeros`, `I`-free helpers `eye` are provided);
  * the reference writes JLD2; here a checkpoint is a NumPy `.npz` with the same content: the point-less group (labels,
    sub-labels, per-cluster statistics / posteriors / parameters / weights / split gate history), the model hyper
    parameters, `iter`, `total_time` and the path of the parameter file -- plus the sampler's RNG state and epoch
    counters, which make a resumed run continue the SAME chain -- same random streams, same decisions; the GPU worker's statistics are
    equal up to Float64 rounding (a fresh context accumulates every sub-cluster once before it derives any from its cache) -- where the
    reference re-seeds.
"""
import json
import os
import time

import numpy as np

from . import priors as _priors

DEFAULTS = dict(                      # src/global_params.jl
    data_path="", data_prefix="", iterations=100, hard_clustering=False, initial_clusters=1,
    argmax_sample_stop=5, split_stop=5, random_seed=None, max_split_iter=20, burnout_period=20,
    max_clusters=np.inf, alpha=10.0, hyper_params=None, outlier_mod=0, outlier_hyper_params=None,
    enable_saving=True, model_save_interval=1000, save_path="./", overwrite_prec=False,
    save_file_prefix="checkpoint_", smart_splits=False)


def load_data(path, prefix="", swapDimension=True, mmap=False):
    """utils.jl:5-14.  Returns Dimensions x Samples (a transposed view) with NaN replaced by 0; with
    `swapDimension=False` the array as stored.  `mmap=True` returns the read-only memory map untouched (no NaN
    pass on the host): the caller hands row blocks to `Worker.upload_points_npy`, which cleans them on the GPU."""
    fn = os.path.join(path, prefix + ".npy") if not str(path).endswith(".npy") else path
    if not os.path.exists(fn):
        fn = str(path) + prefix + ".npy"            # the reference concatenates path * prefix * ".npy"
    if mmap:
        arr = np.load(fn, mmap_mode="r")
        return arr.T if swapDimension else arr
    arr = np.load(fn)
    if np.issubdtype(arr.dtype, np.floating):
        arr = np.where(np.isnan(arr), arr.dtype.type(0), arr)
    return arr.T if swapDimension else arr


class _NS:
    pass


def read_params(model_params):
    """Execute a parameter file and return the recognised settings (defaults of src/global_params.jl).
    The file is EXECUTED (`exec`), exactly as the reference `include`s its Julia parameter file (dp-parallel-sampling.jl:12,181):
    it is code, not data -- only run parameter files you trust."""
    ns = dict(np=np, niw_hyperparams=_priors.niw_hyperparams, multinomial_hyper=_priors.multinomial_hyper,
              Inf=np.inf, nothing=None, true=True, false=False, zeros=np.zeros, ones=np.ones, eye=np.eye)
    mod = _NS()
    mod.niw_hyperparams = _priors.niw_hyperparams
    mod.multinomial_hyper = _priors.multinomial_hyper
    ns["DPMMSubClusters"] = mod
    with open(model_params) as f:
        exec(compile(f.read(), model_params, "exec"), ns)
    out = dict(DEFAULTS)
    for k in DEFAULTS:
        if k in ns:
            out[k] = ns[k]
    if "α" in ns:
        out["alpha"] = ns["α"]
    if out["hyper_params"] is None:
        raise ValueError(f"{model_params}: hyper_params is not set")
    return out


# ------------------------------------------------------------------------------------------------ save / load
# cluster state of the native engine that a checkpoint carries (cluster order; posteriors are recomputed from the statistics)
_STATE_FIELDS = ("packed", "lr_weights", "weights", "splittable", "hist", "points_count", "counters")


def _prior_to_dict(prior):
    if prior.kind == _priors.PRIOR_NIW:
        return dict(prior_kind="niw", prior_kappa=prior.kappa, prior_nu=prior.nu, prior_m=prior.m, prior_psi=prior.psi)
    return dict(prior_kind="multinomial", prior_alpha=np.asarray(prior.alpha))


def _prior_from_dict(d):
    if str(d["prior_kind"]) == "niw":
        return _priors.niw_hyperparams(float(d["prior_kappa"]), d["prior_m"], float(d["prior_nu"]), d["prior_psi"])
    return _priors.multinomial_hyper(d["prior_alpha"])


def checkpoint_filename(path, prefix, it):
    return f"{path}{prefix}_{it}.npz"                 # path * filename * "_" * iter (dp-parallel-sampling.jl:451)


def save_model(sampler, path, prefix, it, total_time, global_params="none"):
    """Collective (every rank calls it): labels are gathered, rank 0 writes the file and returns its name.  The file holds
    what the reference's pts_less_group holds (dp-parallel-sampling.jl:450-455, ds.jl:60-66): labels, sub-labels and the
    cluster state -- here the packed statistics of every (cluster, sub-cluster), the burn-in histories, weights and the
    epoch counters of the counter-based RNG, so that a resumed run continues the SAME chain."""
    labels, sub = sampler.comm.gather_labels(sampler.wk)
    if sampler.comm.rank != 0:
        return None
    m = sampler.model
    d = dict(format="dpmm-checkpoint-2", iter=int(it), total_time=float(total_time), global_params=str(global_params),
             labels=labels.astype(np.int64), labels_subcluster=sub.astype(np.int64), K=int(sampler.K),
             alpha=float(sampler.alpha), total_dim=int(sampler.n_total), seed=np.uint64(sampler.seed),
             burnout=int(sampler.burnout), argmax_sample_stop=int(sampler.argmax_sample_stop), split_stop=int(sampler.split_stop),
             smart_splits=bool(sampler.smart_splits), max_split_iter=int(sampler.max_split_iter), hard_clustering=bool(sampler.hard_clustering))
    d.update(_prior_to_dict(sampler.prior))
    if sampler.outlier_weight > 0:
        d["outlier_weight"] = float(sampler.outlier_weight)
        d.update({"out_" + k: v for k, v in _prior_to_dict(sampler.outlier_prior).items()})
    for k in _STATE_FIELDS:
        d["s_" + k] = m.get(k)
    for k, v in sampler.params.items():
        d["par_" + k] = np.asarray(v)
    fn = checkpoint_filename(path, prefix, it)
    os.makedirs(os.path.dirname(os.path.abspath(fn)), exist_ok=True)
    tmp = fn + ".tmp.npz"
    np.savez(tmp, **d)
    os.replace(tmp, fn)
    return fn


def load_checkpoint(filename):
    with np.load(filename, allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    if str(d.get("format")) != "dpmm-checkpoint-2":
        raise ValueError(f"{filename} is not a checkpoint of this package (format {d.get('format')})")
    return d


def restore_sampler(sampler, ck):
    """Put a freshly built sampler (points uploaded, no clusters yet) into the saved state.  `ck` from load_checkpoint."""
    K = int(ck["K"])
    if int(ck["burnout"]) != sampler.burnout:
        # the burn-in histories are burnout+5 wide and the gate averages over exactly `burnout` entries (shared_actions.jl:51-63)
        raise ValueError(f"checkpoint was written with burnout={int(ck['burnout'])}; resuming with burnout={sampler.burnout} is not supported")
    for k in ("smart_splits", "hard_clustering"):
        if k in ck:
            setattr(sampler, k, bool(ck[k]))
    for k in ("max_split_iter", "argmax_sample_stop", "split_stop"):
        if k in ck:
            setattr(sampler, k, int(ck[k]))
    if "outlier_weight" in ck:
        sampler.outlier_weight = float(ck["outlier_weight"])
        sampler.outlier_prior = _prior_from_dict({k[4:]: v for k, v in ck.items() if k.startswith("out_")})
    sampler._configure()
    lo = getattr(sampler.wk, "first_index", 0)
    n = sampler.wk.n
    sampler.wk.set_labels(ck["labels"][lo:lo + n], ck["labels_subcluster"][lo:lo + n])
    sampler.wk.set_num_clusters(K)
    m = sampler.model
    m.set("K", K)
    for k in _STATE_FIELDS:
        m.set(k, ck["s_" + k])
    for k, v in ck.items():
        if k.startswith("par_"):
            m.set(k[4:], v)
    return sampler


class SaveHook:
    """run_model's `i % model_save_interval == 0 && should_save_model` hook (dp-parallel-sampling.jl:398-403)."""

    def __init__(self, path, prefix, interval, global_params="none", prev_time=0.0, verbose=False):
        self.path, self.prefix, self.interval = path, prefix, int(interval)
        self.global_params, self.verbose = global_params, verbose
        self.start = time.perf_counter() - prev_time
        self.files = []

    def __call__(self, i, sampler):
        if self.interval > 0 and i % self.interval == 0:
            t0 = time.perf_counter()
            fn = save_model(sampler, self.path, self.prefix, i, time.perf_counter() - self.start, self.global_params)
            if fn:
                self.files.append(fn)
                if self.verbose:
                    print(f"Saving Model:\n  {time.perf_counter() - t0:.6f} seconds -> {fn}")
