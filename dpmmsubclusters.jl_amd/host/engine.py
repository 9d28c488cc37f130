"""ctypes binding of include/dpmm_host.h: the native master half of the sweep (libdpmmhost.so, host/csrc/dpmm_model.cpp).

`Model` wraps a `dpmmh_model`; `native_worker_table` fills the model's worker table with the ADDRESSES of the libdpmmhip.so
entry points (no Python between the master's maths and the kernels: one `dpmmh_group_step` call is one sweep);
`python_worker_table` wraps any Python object with the same methods (the oracle-backed test worker of the CPU tests).
"""
import ctypes

import numpy as np

from . import native

_vp = ctypes.c_void_p
_i64p = ctypes.POINTER(ctypes.c_int64)
_pp = ctypes.POINTER(ctypes.c_void_p)

F_STAGING = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, _pp, _pp, _pp, _pp, _pp, _pp)
F_INT = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int)
F_SWEEP = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint32, ctypes.c_int)
F_STEP_STATS = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint32, _pp, _pp)
F_STATS = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _i64p, ctypes.c_int, _pp)
F_M_SETUP = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_double, ctypes.c_double, _vp, _vp)
F_M_STEP = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint32, _pp)
F_M_STEPM = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint32, _vp, ctypes.c_uint32, _pp, _pp)
F_M_STATS = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _i64p, ctypes.c_int)
F_M_POST = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _i64p, _vp, ctypes.c_int, _pp)
F_M_DRAW = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint32, ctypes.c_int, _vp, _vp, _vp)
F_M_PAIRS = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, _pp)
F_M_PAIRSA = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, _vp, ctypes.c_int)
F_M_PUT = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, ctypes.c_int)
F_M_ROWS = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, ctypes.c_int, _vp)
F_M_DRAWS = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp)
F_MM_SETUP = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, _vp)
F_MM_DRAW = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, _vp, _vp)
F_MM_DRAWS = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, _vp)
F_MM_PAIRSA = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int)
F_MM_MARG = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, _pp, _pp, ctypes.POINTER(ctypes.c_int))
F_MM_ROWSD = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int)
F_MM_ROWSW = ctypes.CFUNCTYPE(ctypes.c_int, _vp)
F_SPLIT = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _i64p, _i64p, ctypes.c_int, ctypes.c_uint32)
F_MERGE = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _i64p, _i64p, ctypes.c_int)
F_REMOVE = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _i64p, ctypes.c_int)
F_RESET = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _i64p, ctypes.c_int, ctypes.c_uint32)
F_INIT = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint32)
F_GATHER = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, ctypes.c_int64, _vp)
F_ERR = ctypes.CFUNCTYPE(ctypes.c_char_p, _vp)
F_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _i64p, ctypes.c_int)


class WorkerTable(ctypes.Structure):
    """struct dpmmh_worker (include/dpmm_host.h)."""
    _fields_ = [("ctx", _vp), ("rank", ctypes.c_int), ("world", ctypes.c_int),
                ("params_staging", F_STAGING), ("commit_params", F_INT), ("set_num_clusters", F_INT), ("sweep", F_SWEEP),
                ("step_stats", F_STEP_STATS), ("stats", F_STATS), ("split", F_SPLIT), ("merge", F_MERGE),
                ("remove_empty", F_REMOVE), ("reset_sublabels", F_RESET), ("init_labels", F_INIT), ("allgather", F_GATHER),
                ("last_error", F_ERR),
                ("niw_master_setup", F_M_SETUP), ("step_stats_device", F_M_STEP), ("step_master_device", F_M_STEPM), ("stats_device", F_M_STATS), ("niw_posterior", F_M_POST),
                ("niw_draw", F_M_DRAW), ("niw_pairs", F_M_PAIRS), ("niw_pairs_ahead", F_M_PAIRSA), ("niw_put_rows", F_M_PUT), ("niw_rows", F_M_ROWS), ("niw_draws", F_M_DRAWS),
                ("mult_master_setup", F_MM_SETUP), ("mult_draw", F_MM_DRAW), ("mult_draws", F_MM_DRAWS), ("mult_put_rows", F_M_PUT),
                ("mult_pairs_ahead", F_MM_PAIRSA), ("mult_marginals", F_MM_MARG),
                ("mult_rows_on_demand", F_MM_ROWSD), ("mult_rows_wait", F_MM_ROWSW)]


_NATIVE_MAP = [("params_staging", "dpmm_params_staging", F_STAGING), ("commit_params", "dpmm_commit_params", F_INT),
               ("set_num_clusters", "dpmm_set_num_clusters", F_INT), ("sweep", "dpmm_sweep", F_SWEEP),
               ("step_stats", "dpmm_step_stats", F_STEP_STATS), ("stats", "dpmm_suffstats_host", F_STATS),
               ("split", "dpmm_split", F_SPLIT), ("merge", "dpmm_merge", F_MERGE), ("remove_empty", "dpmm_remove_empty", F_REMOVE),
               ("reset_sublabels", "dpmm_reset_sublabels", F_RESET), ("init_labels", "dpmm_init_labels_from", F_INIT),
               ("allgather", "dpmm_comm_allgather_host", F_GATHER), ("last_error", "dpmm_last_error", F_ERR),
               ("niw_master_setup", "dpmm_niw_master_setup", F_M_SETUP), ("step_stats_device", "dpmm_step_stats_device", F_M_STEP), ("step_master_device", "dpmm_step_master_device", F_M_STEPM),
               ("stats_device", "dpmm_suffstats_device", F_M_STATS), ("niw_posterior", "dpmm_niw_master_posterior", F_M_POST),
               ("niw_draw", "dpmm_niw_master_draw", F_M_DRAW), ("niw_pairs", "dpmm_niw_master_pairs", F_M_PAIRS), ("niw_pairs_ahead", "dpmm_niw_master_pairs_ahead", F_M_PAIRSA),
               ("niw_put_rows", "dpmm_niw_master_put_rows", F_M_PUT),
               ("niw_rows", "dpmm_niw_master_rows", F_M_ROWS), ("niw_draws", "dpmm_niw_master_draws", F_M_DRAWS),
               ("mult_master_setup", "dpmm_mult_master_setup", F_MM_SETUP), ("mult_draw", "dpmm_mult_master_draw", F_MM_DRAW),
               ("mult_draws", "dpmm_mult_master_draws", F_MM_DRAWS), ("mult_put_rows", "dpmm_mult_master_put_rows", F_M_PUT),
               ("mult_pairs_ahead", "dpmm_mult_master_pairs_ahead", F_MM_PAIRSA), ("mult_marginals", "dpmm_mult_master_marginals", F_MM_MARG),
               ("mult_rows_on_demand", "dpmm_mult_master_rows_on_demand", F_MM_ROWSD), ("mult_rows_wait", "dpmm_mult_master_rows_wait", F_MM_ROWSW)]


def native_worker_table(worker, rank=0, world=1):
    """Worker table whose entries are the libdpmmhip.so functions themselves (worker: binding.Worker)."""
    t = WorkerTable()
    t.ctx = worker._h.value
    t.rank, t.world = rank, world
    for field, sym, ftype in _NATIVE_MAP:
        setattr(t, field, ctypes.cast(getattr(worker._lib, sym), ftype))
    return t, [worker]


def python_worker_table(worker, comm=None):
    """Worker table over a Python object (test stand-in).  The object provides params_staging(slots) -> dict of persistent
    numpy arrays (mu, mat, logdet, lr, w, slot), commit_params(K), set_num_clusters, sweep, step_stats(epoch) ->
    (packed, bad), stats(idx or None) -> packed, split, merge, remove_empty, reset_sublabels, init_labels.  `comm`
    (optional) sums the statistics over the ranks (allreduce_np) and gathers host buffers (allgather_bytes)."""
    keep = {"err": b"", "bufs": {}}

    def guard(fn):
        def wrapped(*a):
            try:
                r = fn(*a)
                return 0 if r is None else int(r)
            except Exception as e:  # noqa: BLE001 -- the error text crosses the C boundary through last_error
                import traceback
                keep["err"] = (f"{type(e).__name__}: {e}\n" + traceback.format_exc()).encode()
                return -1
        return wrapped

    def _arr(p, n):
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n > 0 else np.zeros(0, np.int64)

    def _reduce(packed):
        packed = np.ascontiguousarray(packed, np.float64)
        if comm is not None and getattr(comm, "world", 1) > 1:
            packed = comm.allreduce_np(packed)
        return packed

    def staging(_, slots, mu, mat, logdet, lr, w, slot):
        d = worker.params_staging(int(slots))
        keep["bufs"]["staging"] = d
        for ptr, key in ((mu, "mu"), (mat, "mat"), (logdet, "logdet"), (lr, "lr"), (w, "w"), (slot, "slot")):
            a = d.get(key)
            ptr[0] = a.ctypes.data if a is not None else None

    def step_stats(_, epoch, packed_out, bad_out):
        if comm is not None and getattr(comm, "world", 1) > 1:
            counts = comm.allreduce_np(np.ascontiguousarray(worker.bin_counts(), np.float64))
            packed, bad = worker.step_stats(int(epoch), global_counts=counts)
        else:
            packed, bad = worker.step_stats(int(epoch))
        packed = _reduce(packed)
        bad = np.ascontiguousarray(bad, np.uint8)
        keep["bufs"]["packed"], keep["bufs"]["bad"] = packed, bad
        packed_out[0] = packed.ctypes.data
        bad_out[0] = bad.ctypes.data

    def stats(_, idx, n, packed_out):
        packed = _reduce(worker.suffstats_packed(_arr(idx, n) if idx else None))
        keep["bufs"]["packed"] = packed
        packed_out[0] = packed.ctypes.data

    def gather(_, mine, nbytes, out):
        buf = ctypes.string_at(mine, nbytes)
        parts = comm.allgather_bytes(buf) if comm is not None and getattr(comm, "world", 1) > 1 else [buf]
        ctypes.memmove(out, b"".join(parts), nbytes * len(parts))

    t = WorkerTable()
    t.ctx = None
    t.rank, t.world = (getattr(comm, "rank", 0), getattr(comm, "world", 1)) if comm is not None else (0, 1)
    cbs = dict(
        params_staging=F_STAGING(guard(staging)),
        commit_params=F_INT(guard(lambda _, K: worker.commit_params(int(K)))),
        set_num_clusters=F_INT(guard(lambda _, K: worker.set_num_clusters(int(K)))),
        sweep=F_SWEEP(guard(lambda _, ep, fin: worker.sweep(int(ep), bool(fin)))),
        step_stats=F_STEP_STATS(guard(step_stats)),
        stats=F_STATS(guard(stats)),
        split=F_SPLIT(guard(lambda _, a, b, n, ep: worker.split(_arr(a, n), _arr(b, n), int(ep)))),
        merge=F_MERGE(guard(lambda _, a, b, n: worker.merge(_arr(a, n), _arr(b, n)))),
        remove_empty=F_REMOVE(guard(lambda _, pc, K: worker.remove_empty(_arr(pc, K)))),
        reset_sublabels=F_RESET(guard(lambda _, idx, n, ep: worker.reset_sublabels(_arr(idx, n) if idx else None, int(ep)))),
        init_labels=F_INIT(guard(lambda _, ic, first, ep: worker.init_labels_from(int(ic), int(first), int(ep)))),
        allgather=F_GATHER(guard(gather)),
        last_error=F_ERR(lambda _: keep["err"]),
    )
    for k, v in cbs.items():
        setattr(t, k, v)
    return t, [worker, keep, cbs]


OPT_HARD_CLUSTERING, OPT_F32_QUIRK, OPT_THREADS, OPT_SHARE_WORK, OPT_SPIN_US, OPT_PREWAKE, OPT_NUMA_NODE, OPT_DEVICE_MASTER, OPT_DRAW_AHEAD = 1, 2, 3, 4, 5, 6, 7, 8, 9

_FIELDS = {  # name -> (dtype, trailing shape as a function of (K, D, hist_len, stride), rows factor)
    "N": (np.float64, lambda K, D, H, S: (3 * K,)), "sums": (np.float64, lambda K, D, H, S: (3 * K, D)),
    "S": (np.float64, lambda K, D, H, S: (3 * K, D, D)), "packed": (np.float64, lambda K, D, H, S: (2 * K, S)),
    "kappa": (np.float64, lambda K, D, H, S: (3 * K,)), "nu": (np.float64, lambda K, D, H, S: (3 * K,)),
    "logdet_psi": (np.float64, lambda K, D, H, S: (3 * K,)), "log_marginal": (np.float64, lambda K, D, H, S: (3 * K,)),
    "m": (np.float64, lambda K, D, H, S: (3 * K, D)), "U": (np.float64, lambda K, D, H, S: (3 * K, D, D)),
    "alpha_post": (np.float32, lambda K, D, H, S: (3 * K, D)),
    "mu": (np.float32, lambda K, D, H, S: (3 * K, D)), "R": (np.float32, lambda K, D, H, S: (3 * K, D, D)),
    "logdet": (np.float32, lambda K, D, H, S: (3 * K,)), "logp": (np.float32, lambda K, D, H, S: (3 * K, D)),
    "lr_weights": (np.float32, lambda K, D, H, S: (K, 2)), "weights": (np.float32, lambda K, D, H, S: (K,)),
    "splittable": (np.uint8, lambda K, D, H, S: (K,)), "hist": (np.float32, lambda K, D, H, S: (K, H)),
    "points_count": (np.int64, lambda K, D, H, S: (K,)), "counters": (np.int64, lambda K, D, H, S: (8,)),
    "timers": (np.float64, lambda K, D, H, S: (16,)),
}


class EngineError(RuntimeError):
    pass


class Model:
    """dpmmh_model: cluster state + the master half of group_step, in native code."""

    def __init__(self, prior_kind, D, alpha, n_total, seed, burnout, nthreads):
        self._lib = L = native.lib()
        L.dpmmh_model_create.argtypes = [_pp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
        L.dpmmh_model_destroy.argtypes = [_vp]; L.dpmmh_model_destroy.restype = None
        L.dpmmh_model_last_error.argtypes = [_vp]; L.dpmmh_model_last_error.restype = ctypes.c_char_p
        L.dpmmh_model_set_prior_niw.argtypes = [_vp, ctypes.c_int, ctypes.c_double, _vp, ctypes.c_double, _vp]
        L.dpmmh_model_set_prior_mult.argtypes = [_vp, ctypes.c_int, _vp]
        L.dpmmh_model_set_outlier.argtypes = [_vp, ctypes.c_double]
        L.dpmmh_model_set_option.argtypes = [_vp, ctypes.c_int, ctypes.c_double]
        L.dpmmh_model_bind_worker.argtypes = [_vp, ctypes.POINTER(WorkerTable)]
        L.dpmmh_model_set_split_hook.argtypes = [_vp, F_HOOK, _vp]
        L.dpmmh_model_init_first_clusters.argtypes = [_vp, ctypes.c_int]
        L.dpmmh_model_start_from_labels.argtypes = [_vp, ctypes.c_int]
        L.dpmmh_group_step.argtypes = [_vp, ctypes.c_int, ctypes.c_int]
        L.dpmmh_sample_clusters.argtypes = [_vp]
        L.dpmmh_update_suff_stats_posterior.argtypes = [_vp, _i64p, ctypes.c_int]
        L.dpmmh_log_posterior.argtypes = [_vp]; L.dpmmh_log_posterior.restype = ctypes.c_double
        L.dpmmh_model_get.argtypes = [_vp, ctypes.c_char_p, _vp, ctypes.c_int64]; L.dpmmh_model_get.restype = ctypes.c_int64
        L.dpmmh_model_set.argtypes = [_vp, ctypes.c_char_p, _vp, ctypes.c_int64]
        L.dpmmh_timer_names.restype = ctypes.c_char_p
        L.dpmmh_debug_split_log_hr.argtypes = [_vp, _vp]
        L.dpmmh_debug_merge_log_hr.argtypes = [_vp, _vp]
        self._h = _vp()
        if L.dpmmh_model_create(ctypes.byref(self._h), int(prior_kind), int(D), float(alpha), int(n_total), ctypes.c_uint64(int(seed)),
                                int(burnout), int(nthreads)) != 0:
            raise EngineError("dpmmh_model_create failed (bad arguments)")
        self.D, self.kind, self.hist_len = int(D), int(prior_kind), int(burnout) + 5
        self.stride = 1 + self.D + (self.D * (self.D + 1) // 2 if self.kind == 0 else 0)
        self._keep = []
        self._hook = None

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(self._lib.dpmmh_model_last_error(self._h).decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.dpmmh_model_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- configuration
    def set_prior_niw(self, which, kappa, m, nu, psi):
        m = np.ascontiguousarray(m, np.float64); psi = np.ascontiguousarray(psi, np.float64)
        self._chk(self._lib.dpmmh_model_set_prior_niw(self._h, which, float(kappa), m.ctypes.data, float(nu), psi.ctypes.data))

    def set_prior_mult(self, which, alpha):
        a = np.ascontiguousarray(alpha, np.float32)
        self._chk(self._lib.dpmmh_model_set_prior_mult(self._h, which, a.ctypes.data))

    def set_outlier(self, w):
        self._chk(self._lib.dpmmh_model_set_outlier(self._h, float(w)))

    def set_option(self, opt, value):
        self._chk(self._lib.dpmmh_model_set_option(self._h, int(opt), float(value)))

    def bind_worker(self, table, keep):
        self._table = table
        self._keep = keep
        self._chk(self._lib.dpmmh_model_bind_worker(self._h, ctypes.byref(table)))

    def set_split_hook(self, fn):
        """fn(clusters_1based: np.ndarray) or None."""
        if fn is None:
            self._hook = None
            self._chk(self._lib.dpmmh_model_set_split_hook(self._h, ctypes.cast(None, F_HOOK), None))
            return
        err = self

        def cb(_, ids, n):
            try:
                fn(np.ctypeslib.as_array(ids, shape=(n,)).copy())
                return 0
            except Exception as e:  # noqa: BLE001
                err._hook_error = e
                return -1
        self._hook = F_HOOK(cb)
        self._chk(self._lib.dpmmh_model_set_split_hook(self._h, self._hook, None))

    # ---- the sweep
    def init_first_clusters(self, init_clusters):
        self._chk(self._lib.dpmmh_model_init_first_clusters(self._h, int(init_clusters)))

    def start_from_labels(self, K):
        self._chk(self._lib.dpmmh_model_start_from_labels(self._h, int(K)))

    def group_step(self, no_more_splits, final):
        self._chk(self._lib.dpmmh_group_step(self._h, int(bool(no_more_splits)), int(bool(final))))

    def sample_clusters(self):
        self._chk(self._lib.dpmmh_sample_clusters(self._h))

    def update_suff_stats_posterior(self, clusters_1based=None):
        if clusters_1based is None:
            self._chk(self._lib.dpmmh_update_suff_stats_posterior(self._h, None, 0))
        else:
            a = np.ascontiguousarray(clusters_1based, np.int64)
            self._chk(self._lib.dpmmh_update_suff_stats_posterior(self._h, a.ctypes.data_as(_i64p), len(a)))

    def log_posterior(self):
        return float(self._lib.dpmmh_log_posterior(self._h))

    # ---- state
    @property
    def K(self):
        out = np.zeros(1, np.int64)
        self._lib.dpmmh_model_get(self._h, b"K", out.ctypes.data, 8)
        return int(out[0])

    def get(self, field):
        dt, shp = _FIELDS[field]
        shape = shp(self.K, self.D, self.hist_len, self.stride)
        out = np.empty(shape, dt)
        n = self._lib.dpmmh_model_get(self._h, field.encode(), out.ctypes.data, out.nbytes)
        if n != out.nbytes:
            raise EngineError(f"field {field}: " + (self._lib.dpmmh_model_last_error(self._h).decode() if n < 0 else f"size {n} != {out.nbytes}"))
        return out

    def set(self, field, value):
        if field == "K":
            a = np.array([int(value)], np.int64)
        else:
            a = np.ascontiguousarray(value, _FIELDS[field][0])
        self._chk(self._lib.dpmmh_model_set(self._h, field.encode(), a.ctypes.data, a.nbytes))

    def timers(self):
        names = self._lib.dpmmh_timer_names().decode().split(",")
        t = self.get("timers")
        return {n: float(t[i]) for i, n in enumerate(names)}

    def debug_split_log_hr(self):
        out = np.empty(self.K)
        self._chk(self._lib.dpmmh_debug_split_log_hr(self._h, out.ctypes.data))
        return out

    def debug_merge_log_hr(self):
        K = self.K
        out = np.empty((K, K))
        self._chk(self._lib.dpmmh_debug_merge_log_hr(self._h, out.ctypes.data))
        return out
