"""dpmmsubclusters.jl_amd -- MI355X-native worker path of DPMMSubClusters.jl's restricted-Gibbs sweep.

Layout:
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/dpmm_hip.h) -> lib/libdpmmhip.so
  binding.py ctypes binding of the C ABI (`Worker` == one reference worker process == one GPU shard)
  host/      host side of the sampler: priors plug-in surface, posterior draws, split/merge,
             and the `fit` / `dp_parallel` entry points mirroring the reference's API.

The directory name contains a dot, so it is loaded through `__graft_entry__.load_package()`
(importlib, module name `dpmmsubclusters_jl_amd`) rather than a plain import statement.
"""
from .binding import Worker, DpmmError, lib_path, build_library, PRIOR_NIW, PRIOR_MULT  # noqa: F401
