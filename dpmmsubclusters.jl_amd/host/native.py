"""ctypes binding of libdpmmhost.so (host/csrc/dpmm_host.cpp): the native, threaded host-side
maths of the sampler (posterior update + factorisation, NIW / Dirichlet draws)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_f64p = ctypes.POINTER(ctypes.c_double)
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def lib_path():
    return os.path.join(os.path.dirname(_HERE), "lib", "libdpmmhost.so")


def build_library(force=False):
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        args.append("-B")
    from ..binding import locked_make
    locked_make(args, os.path.dirname(lib_path()))
    return lib_path()


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            build_library()
        _LIB = ctypes.CDLL(p)
        _LIB.dpmmh_max_threads.restype = ctypes.c_int
    return _LIB


def _cpu_budget():
    """CPUs this process may actually use: the cgroup CFS quota when there is one (a container can show 256 CPUs
    and grant 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def default_threads():
    env = os.environ.get("DPMM_HOST_THREADS")
    if env:
        return max(1, int(env))
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    # leave room for the HIP runtime's own threads and the Python thread
    return max(1, min(32, (_cpu_budget() - 2 * world) // max(1, world)))


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def niw_posterior(kappa0, nu0, m0, psi0, N, sums, S, want_psi=False, want_U=True, nthreads=None):
    """Batch calc_posterior (priors/niw.jl:20-31) + reverse Cholesky nu*psi = U U'."""
    N = _d(N).ravel(); n = N.size
    D = len(m0)
    sums = _d(sums).reshape(n, D); S = _d(S).reshape(n, D, D)
    kap = np.empty(n); nu = np.empty(n); m = np.empty((n, D)); ld = np.empty(n)
    psi = np.empty((n, D, D)) if want_psi else None
    U = np.empty((n, D, D)) if want_U else None
    m0 = _d(m0); psi0 = _d(psi0)
    lib().dpmmh_niw_posterior(n, D, ctypes.c_double(kappa0), ctypes.c_double(nu0), _p(m0, _f64p), _p(psi0, _f64p),
                              _p(N, _f64p), _p(sums, _f64p), _p(S, _f64p), _p(kap, _f64p), _p(nu, _f64p), _p(m, _f64p),
                              _p(psi, _f64p), _p(U, _f64p), _p(ld, _f64p), nthreads or default_threads())
    return kap, nu, m, psi, U, ld


def niw_logdet_pairs(pairs, kappa0, nu0, m0, psi0, N, sums, S, nthreads=None):
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    N = _d(N).ravel(); n = N.size; D = len(m0)
    sums = _d(sums).reshape(n, D); S = _d(S).reshape(n, D, D)
    out = np.empty(len(pairs))
    m0 = _d(m0); psi0 = _d(psi0)
    lib().dpmmh_niw_logdet_pairs(len(pairs), _p(pairs, _i32p), D, ctypes.c_double(kappa0), ctypes.c_double(nu0),
                                 _p(m0, _f64p), _p(psi0, _f64p), _p(N, _f64p), _p(sums, _f64p), _p(S, _f64p),
                                 _p(out, _f64p), nthreads or default_threads())
    return out


def niw_sample(kappa, nu, m, U, seed, epoch, ids, nthreads=None):
    """sample_distribution (priors/niw.jl:34-40) for a batch: returns mu (f32), R (f32 upper, Sigma^-1 = R'R), logdet Sigma (f32)."""
    kappa = _d(kappa).ravel(); n = kappa.size
    nu = _d(nu).ravel(); m = _d(m); D = m.shape[-1]
    m = m.reshape(n, D); U = _d(U).reshape(n, D, D)
    ids = np.ascontiguousarray(ids, np.int32)
    mu = np.empty((n, D), np.float32); R = np.empty((n, D, D), np.float32); ld = np.empty(n, np.float32)
    lib().dpmmh_niw_sample(n, D, _p(kappa, _f64p), _p(nu, _f64p), _p(m, _f64p), _p(U, _f64p), ctypes.c_uint64(seed),
                           ctypes.c_uint32(epoch), _p(ids, _i32p), _p(mu, _f32p), _p(R, _f32p), _p(ld, _f32p),
                           nthreads or default_threads())
    return mu, R, ld


def niw_noise(n, D, seed, epoch, ids, nthreads=None):
    """Pre-generate the standard-normal noise of `n` draws (see dpmmh_niw_noise)."""
    ids = np.ascontiguousarray(ids, np.int32)
    A = np.empty((n, D, D)); xi = np.empty((n, D))
    lib().dpmmh_niw_noise(n, D, ctypes.c_uint64(seed), ctypes.c_uint32(epoch), _p(ids, _i32p), _p(A, _f64p), _p(xi, _f64p),
                          nthreads or default_threads())
    return A, xi


def niw_sample_noise(kappa, nu, m, U, seed, epoch, ids, A, xi, nthreads=None):
    kappa = _d(kappa).ravel(); n = kappa.size
    nu = _d(nu).ravel(); m = _d(m); D = m.shape[-1]
    m = m.reshape(n, D); U = _d(U).reshape(n, D, D)
    ids = np.ascontiguousarray(ids, np.int32)
    assert A.shape[0] >= n and A.flags.c_contiguous and xi.flags.c_contiguous
    mu = np.empty((n, D), np.float32); R = np.empty((n, D, D), np.float32); ld = np.empty(n, np.float32)
    lib().dpmmh_niw_sample_noise(n, D, _p(kappa, _f64p), _p(nu, _f64p), _p(m, _f64p), _p(U, _f64p), ctypes.c_uint64(seed),
                                 ctypes.c_uint32(epoch), _p(ids, _i32p), _p(A, _f64p), _p(xi, _f64p), _p(mu, _f32p),
                                 _p(R, _f32p), _p(ld, _f32p), nthreads or default_threads())
    return mu, R, ld


def niw_expand(R, want_sigma=True, nthreads=None):
    R = np.ascontiguousarray(R, np.float32)
    n, D = R.shape[0], R.shape[-1]
    inv = np.empty((n, D, D)); sig = np.empty((n, D, D)) if want_sigma else None
    lib().dpmmh_niw_expand(n, D, _p(R, _f32p), _p(inv, _f64p), _p(sig, _f64p), nthreads or default_threads())
    return inv, sig


def dirichlet_log(alpha, seed, epoch, ids, nthreads=None):
    alpha = np.ascontiguousarray(alpha, np.float32)
    n, D = alpha.shape
    ids = np.ascontiguousarray(ids, np.int32)
    out = np.empty((n, D), np.float32)
    lib().dpmmh_dirichlet_log(n, D, _p(alpha, _f32p), ctypes.c_uint64(seed), ctypes.c_uint32(epoch), _p(ids, _i32p),
                              _p(out, _f32p), nthreads or default_threads())
    return out
