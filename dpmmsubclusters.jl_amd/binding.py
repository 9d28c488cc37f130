"""ctypes binding of include/dpmm_hip.h and its companions dpmm_hip_master.h / dpmm_hip_debug.h.  No fallback: if libdpmmhip.so is missing or no
gfx950 device is usable, construction raises -- nothing here computes on the CPU."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PRIOR_NIW, PRIOR_MULT = 0, 1
MAX_CLUSTERS = 1024

_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_f32p = ctypes.POINTER(ctypes.c_float)
_c_f64p = ctypes.POINTER(ctypes.c_double)

# every symbol the three worker headers declare: (name, restype, argtypes)
ABI = [
    ("dpmm_abi_version", ctypes.c_int, []),
    ("dpmm_create", ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64]),
    ("dpmm_destroy", ctypes.c_int, [ctypes.c_void_p]),
    ("dpmm_last_error", ctypes.c_char_p, [ctypes.c_void_p]),
    ("dpmm_upload_points", ctypes.c_int, [ctypes.c_void_p, _c_f32p, ctypes.c_int64]),
    ("dpmm_upload_points_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    ("dpmm_upload_points_npy", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int]),
    ("dpmm_init_labels", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]),
    ("dpmm_set_labels", ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_i64p]),
    ("dpmm_get_labels", ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_i64p]),
    ("dpmm_set_params_niw", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p]),
    ("dpmm_set_params_niw_chol", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p]),
    ("dpmm_set_params_mult", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_f32p, _c_f32p, _c_f32p]),
    ("dpmm_set_num_clusters", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("dpmm_num_clusters", ctypes.c_int, [ctypes.c_void_p]),
    ("dpmm_numa_node", ctypes.c_int, [ctypes.c_void_p]),
    ("dpmm_niw_master_setup", ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    ("dpmm_step_stats_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]),
    ("dpmm_step_master_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    ("dpmm_suffstats_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    ("dpmm_niw_master_posterior", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    ("dpmm_niw_master_draw", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("dpmm_niw_master_pairs_ahead", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    ("dpmm_niw_master_pairs", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    ("dpmm_niw_master_put_rows", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    ("dpmm_niw_master_rows", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    ("dpmm_niw_master_draws", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("dpmm_debug_niw_draw_inputs", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("dpmm_mult_master_setup", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("dpmm_mult_master_draw", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    ("dpmm_mult_master_draws", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    ("dpmm_mult_master_put_rows", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    ("dpmm_mult_master_pairs_ahead", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    ("dpmm_mult_master_marginals", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]),
    ("dpmm_mult_master_rows_on_demand", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("dpmm_mult_master_rows_wait", ctypes.c_int, [ctypes.c_void_p]),
    ("dpmm_sweep", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]),
    ("dpmm_packed_stride", ctypes.c_int64, [ctypes.c_void_p]),
    ("dpmm_suffstats_packed_device", ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int, ctypes.c_void_p]),
    ("dpmm_suffstats_packed", ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int, _c_f64p]),
    ("dpmm_unpack_suffstats", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_f64p, _c_f64p, _c_f64p, _c_f64p]),
    ("dpmm_split", ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_i64p, ctypes.c_int, ctypes.c_uint32]),
    ("dpmm_merge", ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_i64p, ctypes.c_int]),
    ("dpmm_remove_empty", ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int]),
    ("dpmm_reset_sublabels", ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int, ctypes.c_uint32]),
    ("dpmm_set_predictive_niw", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p]),
    ("dpmm_set_predictive_mult", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_f32p, _c_f32p]),
    ("dpmm_predict", ctypes.c_int, [ctypes.c_void_p, _c_f32p]),
    ("dpmm_predict_points", ctypes.c_int, [ctypes.c_void_p, _c_i64p, _c_f32p]),
    ("dpmm_set_ground_truth", ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int]),
    ("dpmm_contingency", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_i64p]),
    ("dpmm_bin_counts", ctypes.c_int, [ctypes.c_void_p, _c_i64p]),
    ("dpmm_smart_project", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _c_f64p, _c_f64p, _c_f64p, _c_i64p]),
    ("dpmm_smart_kmeans_iter", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_double, _c_f64p]),
    ("dpmm_smart_assign", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_double]),
    ("dpmm_debug_loglik", ctypes.c_int, [ctypes.c_void_p, _c_f32p]),
    ("dpmm_sync", ctypes.c_int, [ctypes.c_void_p]),
    ("dpmm_stream", ctypes.c_void_p, [ctypes.c_void_p]),
    ("dpmm_last_sweep_parts_ms", ctypes.c_int, [ctypes.c_void_p, _c_f32p]),
    ("dpmm_last_kernel_ms", ctypes.c_int, [ctypes.c_void_p, _c_f32p, _c_f32p]),
    ("dpmm_debug_counters", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]),
    ("dpmm_debug_set_prelaunch_hook", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("dpmm_init_labels_from", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32]),
    ("dpmm_set_option", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]),
    ("dpmm_params_staging", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.POINTER(ctypes.c_void_p)] * 6),
    ("dpmm_commit_params", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("dpmm_suffstats_host", ctypes.c_int, [ctypes.c_void_p, _c_i64p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    ("dpmm_step_stats", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    ("dpmm_debug_subloglik", ctypes.c_int, [ctypes.c_void_p, _c_f32p]),
    ("dpmm_debug_ref_bracket", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, _c_f32p, _c_f32p]),
    ("dpmm_debug_pair_ball", ctypes.c_int, [ctypes.c_void_p, _c_f32p, _c_f32p]),
    ("dpmm_debug_bracket_big", ctypes.c_int, [ctypes.c_void_p, _c_f32p, ctypes.POINTER(ctypes.c_uint32)]),
    ("dpmm_debug_mult_draws_ahead", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]),
    ("dpmm_last_sweep_work", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]),
    ("dpmm_comm_use_library", ctypes.c_int, [ctypes.c_char_p]),
    ("dpmm_comm_unique_id", ctypes.c_int, [ctypes.c_void_p]),
    ("dpmm_comm_init", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    ("dpmm_comm_init_host", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    ("dpmm_comm_info", ctypes.c_int, [ctypes.c_void_p, _c_i64p]),
    ("dpmm_last_comm_ms", ctypes.c_int, [ctypes.c_void_p, _c_f32p, _c_f32p]),
    ("dpmm_comm_destroy", ctypes.c_int, [ctypes.c_void_p]),
    ("dpmm_comm_allgather_host", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
]

HOST_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int)   # dpmm_host_allreduce_fn

# dpmm_set_option keys (include/dpmm_hip.h)
MASTER_NSCALARS = 8          # DPMM_MASTER_NSCALARS
OPT_SCREEN_MARGIN, OPT_TAIL_SCREEN, OPT_PRESCREEN, OPT_ORDERED_SWEEP, OPT_MULT_FORCE_F32, OPT_STATS_ITEMS, OPT_STATS_GROUPS, OPT_TRACE_SLOW, OPT_LOGLIK_REF_CONST, OPT_WAVE_PRIO, OPT_MULT_NO_U8, OPT_SWEEP_GRID, OPT_SWEEP_QUEUE_ROUNDS, OPT_BALL_SCREEN, OPT_KERNEL_TIMING, OPT_STATS_DERIVE, OPT_NOISE_AHEAD, OPT_REF_BRACKET, OPT_SORT_TILE, OPT_ONE_COLLECTIVE, OPT_BF16_SCREENS, OPT_COMM_TIMEOUT_MS, OPT_DIRECTION_SCREEN, _OPT_RESERVED_24, OPT_MULT_DRAWS_AHEAD, OPT_B3_SUBLABELS, OPT_LEAN_TILES, OPT_MASTER_POLL, OPT_CHAIN_FUSION, OPT_LEAN_DIRECTION, OPT_PAIR_BALL = range(1, 32)


class DpmmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdpmmhip error {code}: {msg}")
        self.code = code


def lib_path():
    return os.path.join(_HERE, "lib", "libdpmmhip.so")


def locked_make(args, lock_dir):
    """Run `make` under an exclusive file lock: the ranks of a fresh multi-process launch all reach the lazy build at once and
    must not write the same .so concurrently."""
    import fcntl
    os.makedirs(lock_dir, exist_ok=True)
    with open(os.path.join(lock_dir, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.check_call(args)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def build_library(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if force:
        args.append("-B")
    locked_make(args, os.path.join(_HERE, "lib"))
    return lib_path()


_LIB = None


ABI_VERSION = 3      # DPMM_ABI_VERSION of include/dpmm_hip.h this file is written against


def load_library():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            try:   # fresh checkout: compile for gfx950 (hipcc cross-compiles without a GPU); never falls back to CPU code
                build_library()
            except Exception as e:
                raise FileNotFoundError(f"{p} not built and building it failed ({e}): run __graft_entry__.build() "
                                        "(there is no CPU fallback)") from e
        # PyTorch ships its own libamdhip64 under the same SONAME.  Whichever copy is mapped first serves BOTH users; if
        # this library pulled in the system copy first, a later `import torch` (multi-GPU runs use torch.distributed)
        # would find no GPUs.  Mapping torch's copy first keeps one consistent HIP runtime in the process.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover  (torch is optional for single-GPU use)
            pass
        lib = ctypes.CDLL(p)
        for name, res, args in ABI:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.dpmm_abi_version() != ABI_VERSION:      # (dpmm_last_sweep_work fills 16 words since version 3: an older library would be handed too long a buffer, a newer one may overflow ours)
            raise ImportError(f"{p}: DPMM_ABI_VERSION {lib.dpmm_abi_version()}, this binding is written for {ABI_VERSION} (include/dpmm_hip.h)")
        _LIB = lib
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Worker:
    """One shard of the points on one GPU: the stand-in for one reference worker process
    (the functions a worker runs on its `localpart`s, src/local_clusters_actions.jl)."""

    def __init__(self, prior, D, n_local, first_index=0, device=0, seed=0, timing=True):
        """timing: record HIP events around the sweep / statistics / all-reduces (last_kernel_ms, last_comm_ms).  On by default for
        this binding (tests, benches, scripts read them); the library's own default -- and what fit / dp_parallel use -- is off:
        each event costs ~5 us between two kernels."""
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.prior, self.D, self.n, self.first_index, self.device = prior, int(D), int(n_local), int(first_index), device
        rc = self._lib.dpmm_create(ctypes.byref(self._h), prior, D, n_local, first_index, device, ctypes.c_uint64(seed))
        if rc != 0:
            raise DpmmError(rc, self._lib.dpmm_last_error(None).decode())
        self.K = 0
        self.packed_stride = int(self._lib.dpmm_packed_stride(self._h))
        self.timing = bool(timing)
        if timing:
            self._chk(self._lib.dpmm_set_option(self._h, OPT_KERNEL_TIMING, 7.0))

    def _chk(self, rc):
        if rc != 0:
            err = DpmmError(rc, self._lib.dpmm_last_error(self._h).decode())
            cause = getattr(self, "_comm_error", None)          # an exception inside the host all-reduce callback (comm_init_host)
            if cause is not None:
                self._comm_error = None
                raise err from cause
            raise err

    @property
    def K(self):
        """Number of clusters the ctx currently knows (the native engine changes it through the C ABI directly)."""
        h = getattr(self, "_h", None)
        return int(self._lib.dpmm_num_clusters(h)) if h is not None and h.value else 0

    @K.setter
    def K(self, value):
        pass

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            try:
                early = self.debug_counters()[0]
            except Exception:
                early = 0
            if early:
                import sys
                print("dpmm worker: %d event wait(s) returned before the posterior records had reached host memory "
                      "(waited out; results unaffected)" % early, file=sys.stderr, flush=True)
            self._lib.dpmm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data
    def upload_points(self, X):
        """X: (n_local, ld) float32 C-contiguous, row i = point i (== Julia's D x n column-major)."""
        X = _f32(X)
        assert X.ndim == 2 and X.shape[0] == self.n and X.shape[1] >= self.D
        self._chk(self._lib.dpmm_upload_points(self._h, _p(X, _c_f32p), X.shape[1]))

    def upload_points_npy(self, rows, nan_to_zero=True):
        """rows: (n_local, >= D) array of a Samples x Dimensions .npy file (Float32 or Float64, C-contiguous; a
        read-only memory map is fine).  Float32 conversion and NaN -> 0 (utils.jl:9-13) happen on the device."""
        rows = np.asarray(rows)
        if rows.dtype not in (np.float32, np.float64):
            rows = rows.astype(np.float32)
        if not rows.flags.c_contiguous:
            rows = np.ascontiguousarray(rows)
        assert rows.ndim == 2 and rows.shape[0] == self.n and rows.shape[1] >= self.D
        self._chk(self._lib.dpmm_upload_points_npy(self._h, ctypes.c_void_p(rows.ctypes.data), int(rows.dtype == np.float64),
                                                   rows.shape[1], int(bool(nan_to_zero))))

    def upload_points_device(self, ptr, ldx):
        self._chk(self._lib.dpmm_upload_points_device(self._h, ctypes.c_void_p(ptr), ldx))

    def init_labels(self, init_clusters, epoch):
        self._chk(self._lib.dpmm_init_labels(self._h, init_clusters, epoch))

    def set_labels(self, labels=None, sub=None):
        labels = _i64(labels) if labels is not None else None
        sub = _i64(sub) if sub is not None else None
        self._chk(self._lib.dpmm_set_labels(self._h, _p(labels, _c_i64p), _p(sub, _c_i64p)))

    def get_labels(self):
        lab = np.empty(self.n, np.int64); sub = np.empty(self.n, np.int64)
        self._chk(self._lib.dpmm_get_labels(self._h, _p(lab, _c_i64p), _p(sub, _c_i64p)))
        return lab, sub

    # ---- parameters
    def set_params_niw(self, mu, inv_sigma, logdet, lr_weights, weights):
        K = len(weights)
        mu, inv_sigma, logdet, lr_weights, weights = map(_f32, (mu, inv_sigma, logdet, lr_weights, weights))
        assert mu.shape == (3 * K, self.D) and inv_sigma.size == 3 * K * self.D * self.D and logdet.shape == (3 * K,) and lr_weights.shape == (K, 2)
        self._chk(self._lib.dpmm_set_params_niw(self._h, K, _p(mu, _c_f32p), _p(inv_sigma, _c_f32p), _p(logdet, _c_f32p), _p(lr_weights, _c_f32p), _p(weights, _c_f32p)))
        self.K = K

    def set_params_niw_chol(self, mu, R, logdet, lr_weights, weights):
        K = len(weights)
        mu, R, logdet, lr_weights, weights = map(_f32, (mu, R, logdet, lr_weights, weights))
        assert mu.shape == (3 * K, self.D) and R.size == 3 * K * self.D * self.D and logdet.shape == (3 * K,) and lr_weights.shape == (K, 2)
        self._chk(self._lib.dpmm_set_params_niw_chol(self._h, K, _p(mu, _c_f32p), _p(R, _c_f32p), _p(logdet, _c_f32p), _p(lr_weights, _c_f32p), _p(weights, _c_f32p)))
        self.K = K

    def set_params_mult(self, logp, lr_weights, weights):
        K = len(weights)
        logp, lr_weights, weights = map(_f32, (logp, lr_weights, weights))
        assert logp.shape == (3 * K, self.D) and lr_weights.shape == (K, 2)
        self._chk(self._lib.dpmm_set_params_mult(self._h, K, _p(logp, _c_f32p), _p(lr_weights, _c_f32p), _p(weights, _c_f32p)))
        self.K = K

    def set_num_clusters(self, K):
        self._chk(self._lib.dpmm_set_num_clusters(self._h, int(K)))
        self.K = int(K)

    # ---- the hot path
    def sweep(self, epoch, final=False):
        self._chk(self._lib.dpmm_sweep(self._h, epoch, int(bool(final))))

    def suffstats_packed(self, cluster_idx=None):
        out = np.empty((2 * self.K, self.packed_stride), np.float64)
        idx = _i64(cluster_idx) if cluster_idx is not None else None
        self._chk(self._lib.dpmm_suffstats_packed(self._h, _p(idx, _c_i64p), 0 if idx is None else len(idx), _p(out, _c_f64p)))
        return out

    def suffstats_packed_device(self, dev_ptr, cluster_idx=None):
        idx = _i64(cluster_idx) if cluster_idx is not None else None
        self._chk(self._lib.dpmm_suffstats_packed_device(self._h, _p(idx, _c_i64p), 0 if idx is None else len(idx), ctypes.c_void_p(dev_ptr)))

    def unpack(self, packed, K=None):
        K = self.K if K is None else K
        packed = np.ascontiguousarray(packed, np.float64)
        N = np.empty((K, 3)); s = np.empty((K, 3, self.D))
        S = np.empty((K, 3, self.D, self.D)) if self.prior == PRIOR_NIW else None
        self._chk(self._lib.dpmm_unpack_suffstats(self._h, K, _p(packed, _c_f64p), _p(N, _c_f64p), _p(s, _c_f64p), _p(S, _c_f64p)))
        return (N, s, S) if S is not None else (N, s)

    def suffstats(self, cluster_idx=None):
        return self.unpack(self.suffstats_packed(cluster_idx))

    # ---- relabel
    def split(self, idx, new_idx, epoch):
        idx, new_idx = _i64(idx), _i64(new_idx)
        self._chk(self._lib.dpmm_split(self._h, _p(idx, _c_i64p), _p(new_idx, _c_i64p), len(idx), epoch))

    def merge(self, idx, new_idx):
        idx, new_idx = _i64(idx), _i64(new_idx)
        self._chk(self._lib.dpmm_merge(self._h, _p(idx, _c_i64p), _p(new_idx, _c_i64p), len(idx)))

    def remove_empty(self, pts_count):
        pc = _i64(pts_count)
        self._chk(self._lib.dpmm_remove_empty(self._h, _p(pc, _c_i64p), len(pc)))

    def reset_sublabels(self, idx, epoch):
        if idx is None:
            self._chk(self._lib.dpmm_reset_sublabels(self._h, None, 0, epoch))
        else:
            idx = _i64(idx)
            self._chk(self._lib.dpmm_reset_sublabels(self._h, _p(idx, _c_i64p), len(idx), epoch))

    # ---- prediction (posterior predictive table)
    def predict_table_niw(self, m, R, logdet, df, weights, points=False):
        K = len(weights)
        m, R, logdet, df, weights = map(_f32, (m, R, logdet, df, weights))
        assert m.shape == (K, self.D) and R.size == K * self.D * self.D
        self._chk(self._lib.dpmm_set_predictive_niw(self._h, K, _p(m, _c_f32p), _p(R, _c_f32p), _p(logdet, _c_f32p), _p(df, _c_f32p), _p(weights, _c_f32p)))
        self.K = K
        if points:
            return self._predict_points(K)
        out = np.empty((K, self.n), np.float32)
        self._chk(self._lib.dpmm_predict(self._h, _p(out, _c_f32p)))
        return out

    supports_predict_points = True

    def _predict_points(self, K):
        lab = np.empty(self.n, np.int64); probs = np.empty((self.n, K), np.float32)
        self._chk(self._lib.dpmm_predict_points(self._h, _p(lab, _c_i64p), _p(probs, _c_f32p)))
        return lab, probs

    def predict_table_mult(self, logp, weights, points=False):
        K = len(weights)
        logp, weights = _f32(logp), _f32(weights)
        assert logp.shape == (K, self.D)
        self._chk(self._lib.dpmm_set_predictive_mult(self._h, K, _p(logp, _c_f32p), _p(weights, _c_f32p)))
        self.K = K
        if points:
            return self._predict_points(K)
        out = np.empty((K, self.n), np.float32)
        self._chk(self._lib.dpmm_predict(self._h, _p(out, _c_f32p)))
        return out

    # ---- on-device evaluation
    def set_ground_truth(self, gt):
        """gt: integer ids of this shard's points (any integers; remapped to 0..n_gt-1 by the caller)."""
        gt = _i64(gt)
        self.n_gt = int(gt.max()) + 1 if len(gt) else 1
        self._chk(self._lib.dpmm_set_ground_truth(self._h, _p(gt, _c_i64p), self.n_gt))

    def set_ground_truth_range(self, gt, n_gt):
        gt = _i64(gt)
        self.n_gt = int(n_gt)
        self._chk(self._lib.dpmm_set_ground_truth(self._h, _p(gt, _c_i64p), self.n_gt))

    def contingency(self, K=None):
        K = self.K if K is None else K
        out = np.zeros((K, self.n_gt), np.int64)
        self._chk(self._lib.dpmm_contingency(self._h, K, _p(out, _c_i64p)))
        return out

    def bin_counts(self):
        """(K, 2) Int64: points per (cluster, sub-cluster) of this shard -- the N of the l / r statistics."""
        out = np.zeros((self.K, 2), np.int64)
        self._chk(self._lib.dpmm_bin_counts(self._h, _p(out, _c_i64p)))
        return out

    # ---- smart splits (worker halves of smart_cluster_init!)
    def smart_project(self, cluster, v, mu):
        """Projections t = v.(x - mu) of the points of `cluster` (1-based), in arbitrary order (Float64)."""
        v = np.ascontiguousarray(v, np.float64); mu = np.ascontiguousarray(mu, np.float64)
        assert v.shape == (self.D,) and mu.shape == (self.D,)
        vals = np.empty(max(self.n, 1), np.float64)
        cnt = np.zeros(1, np.int64)
        self._chk(self._lib.dpmm_smart_project(self._h, int(cluster), _p(v, _c_f64p), _p(mu, _c_f64p), _p(vals, _c_f64p), _p(cnt, _c_i64p)))
        return vals[:int(cnt[0])]

    def smart_kmeans_iter(self, cluster, m_lo, m_hi):
        out = np.zeros(4, np.float64)
        self._chk(self._lib.dpmm_smart_kmeans_iter(self._h, int(cluster), float(m_lo), float(m_hi), _p(out, _c_f64p)))
        return out

    def smart_assign(self, cluster, m_lo, m_hi):
        self._chk(self._lib.dpmm_smart_assign(self._h, int(cluster), float(m_lo), float(m_hi)))

    # ---- options / collective
    def set_option(self, option, value):
        self._chk(self._lib.dpmm_set_option(self._h, int(option), float(value)))

    def _use_torch_rccl(self):
        """One RCCL per process: when torch is importable, bind the library to torch's own librccl.so."""
        try:
            import torch
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(cand):
                self._lib.dpmm_comm_use_library(cand.encode())
        except ImportError:
            pass

    def comm_unique_id(self):
        self._use_torch_rccl()
        buf = ctypes.create_string_buffer(128)
        rc = self._lib.dpmm_comm_unique_id(buf)
        if rc != 0:
            raise DpmmError(rc, self._lib.dpmm_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        self._use_torch_rccl()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._lib.dpmm_comm_init(self._h, buf, int(rank), int(world)))

    def comm_init_host(self, rank, world, allreduce):
        """dpmm_comm_init_host: `allreduce(arr)` sums a 1-D numpy array (float64 or int64, a VIEW of the library's pinned staging)
        over the ranks IN PLACE.  The statistics calls then return rows summed over the ranks, as with comm_init."""
        def cb(_, buf, count, is_f64):
            try:
                t = ctypes.c_double if is_f64 else ctypes.c_int64
                allreduce(np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(t)), shape=(int(count),)))
                return 0
            except Exception as e:  # noqa: BLE001 -- must not unwind through the C frames; the library reports DPMM_ECOMM
                self._comm_error = e
                return 1
        self._host_cb = HOST_ALLREDUCE_FN(cb)       # keep the trampoline alive as long as the context
        self._chk(self._lib.dpmm_comm_init_host(self._h, int(rank), int(world), ctypes.cast(self._host_cb, ctypes.c_void_p), None))

    def comm_info(self):
        out = np.zeros(8, np.int64)
        self._chk(self._lib.dpmm_comm_info(self._h, _p(out, _c_i64p)))
        return dict(world=int(out[0]), rank=int(out[1]), transport={0: "none", 1: "rccl", 2: "host"}[int(out[2])],
                    counts_bytes=int(out[3]), rows_bytes=int(out[4]), allreduces=int(out[5]), one_collective=bool(out[7]))

    def last_comm_ms(self):
        """(occupancy all-reduce ms, packed-row all-reduce ms) of the last statistics pass (HIP events on the ctx stream)."""
        a = ctypes.c_float(); b = ctypes.c_float()
        self._chk(self._lib.dpmm_last_comm_ms(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def numa_node(self):
        """dpmm_numa_node: NUMA node of the host this context's GPU is attached to (-1: unknown)."""
        return int(self._lib.dpmm_numa_node(self._h))

    # ---- the master's dense maths on the device (NIW)
    def master_setup(self, kappa, nu, m, psi):
        m = np.ascontiguousarray(m, np.float64); psi = np.ascontiguousarray(psi, np.float64)
        self._chk(self._lib.dpmm_niw_master_setup(self._h, float(kappa), float(nu), m.ctypes.data, psi.ctypes.data))

    def step_stats_device(self, reset_epoch):
        bad = ctypes.c_void_p()
        self._chk(self._lib.dpmm_step_stats_device(self._h, int(reset_epoch), ctypes.byref(bad)))
        return np.ctypeslib.as_array(ctypes.cast(bad, ctypes.POINTER(ctypes.c_uint8)), shape=(self.K,)).copy()

    def suffstats_device(self, cluster_idx=None):
        idx = None if cluster_idx is None else np.ascontiguousarray(cluster_idx, np.int64)
        self._chk(self._lib.dpmm_suffstats_device(self._h, None if idx is None else idx.ctypes.data, 0 if idx is None else len(idx)))

    def master_posterior(self, clusters, slots):
        """dpmm_niw_master_posterior: (n, 3, 8) float64 {N, kappa', nu', log det(nu' psi'), log Gamma_D(nu' / 2), 3 spare} for the listed clusters (1-based)."""
        slots = np.ascontiguousarray(slots, np.int32)
        cl = None if clusters is None else np.ascontiguousarray(clusters, np.int64)
        out = ctypes.c_void_p()
        self._chk(self._lib.dpmm_niw_master_posterior(self._h, None if cl is None else cl.ctypes.data, slots.ctypes.data, len(slots), ctypes.byref(out)))
        return np.ctypeslib.as_array(ctypes.cast(out, _c_f64p), shape=(len(slots), 3, MASTER_NSCALARS)).copy()

    def master_draw(self, epoch, slot_of_cluster, lr_weights, weights):
        sl = np.ascontiguousarray(slot_of_cluster, np.int32)
        lr = np.ascontiguousarray(lr_weights, np.float32); w = np.ascontiguousarray(weights, np.float32)
        self._chk(self._lib.dpmm_niw_master_draw(self._h, int(epoch), len(sl), sl.ctypes.data, lr.ctypes.data, w.ctypes.data))

    def master_pairs(self, slots_i, slots_j):
        """dpmm_niw_master_pairs: (n, 8) float64 {N, kappa', nu', log det(nu' psi'), log Gamma_D(nu' / 2), 3 spare} of the pooled statistics of the slot pairs."""
        a = np.ascontiguousarray(slots_i, np.int32); b = np.ascontiguousarray(slots_j, np.int32)
        out = ctypes.c_void_p()
        self._chk(self._lib.dpmm_niw_master_pairs(self._h, a.ctypes.data, b.ctypes.data, len(a), ctypes.byref(out)))
        return np.ctypeslib.as_array(ctypes.cast(out, _c_f64p), shape=(len(a), MASTER_NSCALARS)).copy()

    def master_pairs_ahead(self, slots_i, slots_j):
        """dpmm_niw_master_pairs_ahead: the pairs the next step_master_device computes behind its posteriors."""
        a = np.ascontiguousarray(slots_i, np.int32); b = np.ascontiguousarray(slots_j, np.int32)
        self._chk(self._lib.dpmm_niw_master_pairs_ahead(self._h, a.ctypes.data, b.ctypes.data, len(a)))

    def step_master_device(self, reset_epoch, slots, draw_epoch=0):
        """dpmm_step_master_device: statistics + all posteriors in one call -> (bad flags (K,), scalars (K, 3, 8))."""
        sl = np.ascontiguousarray(slots, np.int32)
        bad = ctypes.c_void_p(); out = ctypes.c_void_p()
        self._chk(self._lib.dpmm_step_master_device(self._h, int(reset_epoch), sl.ctypes.data, int(draw_epoch), ctypes.byref(bad), ctypes.byref(out)))
        return (np.ctypeslib.as_array(ctypes.cast(bad, ctypes.POINTER(ctypes.c_uint8)), shape=(self.K,)).copy(),
                np.ctypeslib.as_array(ctypes.cast(out, _c_f64p), shape=(len(sl), 3, MASTER_NSCALARS)).copy())

    def master_rows(self, slots):
        sl = np.ascontiguousarray(slots, np.int32)
        out = np.empty((len(sl), 2, self.packed_stride), np.float64)
        self._chk(self._lib.dpmm_niw_master_rows(self._h, sl.ctypes.data, len(sl), out.ctypes.data))
        return out

    def master_draws(self, K):
        mu = np.empty((3 * K, self.D), np.float32); R = np.empty((3 * K, self.D, self.D), np.float32); ld = np.empty(3 * K, np.float32)
        self._chk(self._lib.dpmm_niw_master_draws(self._h, int(K), mu.ctypes.data, R.ctypes.data, ld.ctypes.data))
        return mu, R, ld

    # ---- the Multinomial master's draws on the device
    def mult_master_setup(self, alpha, alpha_outlier=None):
        a = np.ascontiguousarray(alpha, np.float32); b = None if alpha_outlier is None else np.ascontiguousarray(alpha_outlier, np.float32)
        self._chk(self._lib.dpmm_mult_master_setup(self._h, a.ctypes.data, None if b is None else b.ctypes.data))

    def mult_master_draw(self, epoch, lr_weights, weights, outlier_first=False):
        lr = np.ascontiguousarray(lr_weights, np.float32); w = np.ascontiguousarray(weights, np.float32)
        self._chk(self._lib.dpmm_mult_master_draw(self._h, int(epoch), len(w), int(bool(outlier_first)), lr.ctypes.data, w.ctypes.data))

    def mult_master_draws(self, K):
        out = np.empty((3 * K, self.D), np.float32)
        self._chk(self._lib.dpmm_mult_master_draws(self._h, int(K), out.ctypes.data))
        return out

    def mult_master_pairs_ahead(self, ki, kj, outlier_first=False):
        a = np.ascontiguousarray(ki, np.int32); b = np.ascontiguousarray(kj, np.int32)
        self._chk(self._lib.dpmm_mult_master_pairs_ahead(self._h, int(bool(outlier_first)), a.ctypes.data, b.ctypes.data, len(a)))

    def mult_master_marginals(self, K):
        """-> (N [3K], log-marginals [3K], pooled log-marginals of the pairs asked for ahead)"""
        pn, pl, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int()
        self._chk(self._lib.dpmm_mult_master_marginals(self._h, int(K), ctypes.byref(pn), ctypes.byref(pl), ctypes.byref(n)))
        nl = np.ctypeslib.as_array(ctypes.cast(pn, ctypes.POINTER(ctypes.c_double)), shape=(3 * K, 2)).copy()
        pairs = np.ctypeslib.as_array(ctypes.cast(pl, ctypes.POINTER(ctypes.c_double)), shape=(max(n.value, 1),)).copy()[:n.value]
        return nl[:, 0], nl[:, 1], pairs

    def mult_master_put_rows(self, rows):
        r = np.ascontiguousarray(rows, np.float64)
        self._chk(self._lib.dpmm_mult_master_put_rows(self._h, r.ctypes.data, r.shape[0] // 2))

    def debug_draw_inputs(self, epoch, slot_of_cluster):
        """dpmm_debug_niw_draw_inputs: (A (3K, D, D) Bartlett factors, xi (3K, D)) the device draw of `epoch` consumes."""
        sl = np.ascontiguousarray(slot_of_cluster, np.int32)
        K = len(sl)
        A = np.empty((3 * K, self.D, self.D), np.float64); xi = np.empty((3 * K, self.D), np.float64)
        self._chk(self._lib.dpmm_debug_niw_draw_inputs(self._h, int(epoch), K, sl.ctypes.data, A.ctypes.data, xi.ctypes.data))
        return A, xi

    def step_stats(self, reset_epoch):
        """dpmm_step_stats: (packed (2K, stride) float64, bad (K,) uint8) -- copies of the ctx's pinned output."""
        pk, bad = ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(self._lib.dpmm_step_stats(self._h, int(reset_epoch), ctypes.byref(pk), ctypes.byref(bad)))
        packed = np.ctypeslib.as_array(ctypes.cast(pk, _c_f64p), shape=(2 * self.K, self.packed_stride)).copy()
        flags = np.ctypeslib.as_array(ctypes.cast(bad, ctypes.POINTER(ctypes.c_uint8)), shape=(self.K,)).copy()
        return packed, flags

    def init_labels_from(self, init_clusters, first_label, epoch):
        self._chk(self._lib.dpmm_init_labels_from(self._h, int(init_clusters), int(first_label), int(epoch)))

    def last_sweep_work(self):
        """dict of the executed-work counters of the NIW sweeps since the previous call, PER LAUNCH (dpmm_last_sweep_work returns their
        totals and the number of launches, and clears them): after every sweep = that sweep's; after a timed loop = its average."""
        out = (ctypes.c_uint64 * 16)()
        self._chk(self._lib.dpmm_last_sweep_work(self._h, out))
        v = [int(x) for x in out]
        n = max(1, v[7])
        sp = 8 * ((self.K + 15) // 16)      # bf16 matrix instructions of one direction screen: two per 16 clusters and point group
        # executed_flops: Float32 matrix work only (= SQ_INSTS_VALU_MFMA_MOPS_F32 x 512); the bf16 work (brackets, screens) beside it
        dirs, b3 = v[15] & 0xFFFFFFFF, v[15] >> 32      # direction screens | bf16 three-plane sub-cluster evaluations (144 bf16 matrix instructions + 4 Float32 row sums each)
        bf16 = v[8] * v[9] + v[11] * v[12] + v[13] * v[14] + dirs * sp + b3 * 144
        return dict(wave_tiles=v[0] / n, full_evals=v[1] / n, screens16=v[2] / n, tail_pairs=v[3] / n, mfma_per_full=v[4], mfma_per_screen=v[5],
                    flops_per_mfma=v[6], executed_flops=(v[1] * v[4] + v[2] * v[5] + 4 * b3) * v[6] / n, launches=v[7],
                    brackets=v[8] / n, bf16_mfma_per_bracket=v[9], bf16_bottom_screens=v[11] / n, bf16_top_screens=v[13] / n,
                    direction_screens=dirs / n, b3_evals=b3 / n,
                    bf16_mfma=bf16 / n, bf16_flops=bf16 * v[10] / n)

    # ---- diagnostics
    def debug_subloglik(self):
        """(2K, n) float32: row 2k+s = log-likelihood under sub-cluster s of cluster k + log lr_weights[k][s]."""
        out = np.empty((2 * self.K, self.n), np.float32)
        self._chk(self._lib.dpmm_debug_subloglik(self._h, _p(out, _c_f32p)))
        return out

    def debug_ref_bracket(self, cluster, c_override=0.0):
        """(q_hi, q): per point, the reference bracket's upper end and the Float32 quadratic form of `cluster` (1-based); include/dpmm_hip_debug.h."""
        qhi = np.empty(self.n, np.float32); q = np.empty(self.n, np.float32)
        self._chk(self._lib.dpmm_debug_ref_bracket(self._h, int(cluster), ctypes.c_float(c_override), _p(qhi, _c_f32p), _p(q, _c_f32p)))
        return qhi, q

    def debug_pair_ball(self):
        """(pd (K, K), sn (K,)): the pair-ball table of the parameters on the device; include/dpmm_hip_debug.h."""
        pd = np.empty((self.K, self.K), np.float32); sn = np.empty(self.K, np.float32)
        self._chk(self._lib.dpmm_debug_pair_ball(self._h, _p(pd, _c_f32p), _p(sn, _c_f32p)))
        return pd, sn

    def debug_bracket_big(self):
        """(aref (n,), tile_flags (ceil(n / 128),)): what the D > 64 sweep reads from the bracket launch in front of it; include/dpmm_hip_debug.h."""
        aref = np.empty(self.n, np.float32); fl = np.empty((self.n + 127) // 128, np.uint32)
        self._chk(self._lib.dpmm_debug_bracket_big(self._h, _p(aref, _c_f32p), fl.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        return aref, fl

    def debug_mult_draws_ahead(self):
        """dpmm_mult_master_draw calls that took the draws launched ahead by dpmm_step_stats; include/dpmm_hip_debug.h."""
        v = ctypes.c_longlong(0)
        self._chk(self._lib.dpmm_debug_mult_draws_ahead(self._h, ctypes.byref(v)))
        return int(v.value)

    def debug_loglik(self):
        out = np.empty((self.K, self.n), np.float32)
        self._chk(self._lib.dpmm_debug_loglik(self._h, _p(out, _c_f32p)))
        return out

    def sync(self):
        self._chk(self._lib.dpmm_sync(self._h))

    @property
    def stream(self):
        return self._lib.dpmm_stream(self._h)

    def set_timing(self, on):
        """on: False / True (sweep + statistics + all-reduce events) or a bit mask 1 (sweep kernel) | 2 (statistics) | 4 (all-reduces)."""
        mask = 7 if on is True else int(on)      # (bit 3 = 8: the three parts of a D <= 64 sweep, last_sweep_parts_ms)
        self.timing = mask != 0
        self.set_option(OPT_KERNEL_TIMING, float(mask))

    def last_sweep_parts_ms(self):
        """(lean, labels, sub-labels) milliseconds of the last sweep's three launches (timing bits 0 and 3; zeros otherwise)."""
        out = np.zeros(3, np.float32)
        self._chk(self._lib.dpmm_last_sweep_parts_ms(self._h, _p(out, _c_f32p)))
        return [float(v) for v in out]

    def last_kernel_ms(self):
        if not self.timing:
            raise RuntimeError("kernel timing is off for this Worker (timing=False / set_timing(False)): no events were recorded")
        a = ctypes.c_float(); b = ctypes.c_float()
        self._chk(self._lib.dpmm_last_kernel_ms(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def debug_counters(self):
        """Health counters of the worker (include/dpmm_hip.h dpmm_debug_counters): [0] = event waits that returned early."""
        out = (ctypes.c_int64 * 4)()
        self._chk(self._lib.dpmm_debug_counters(self._h, out, 4))
        return list(out)
