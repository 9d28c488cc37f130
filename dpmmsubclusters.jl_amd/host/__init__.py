"""Host side of the sampler (above the C ABI): prior plug-ins, master-side sweep, user API."""
try:  # numpy's BLAS worker threads spin-wait and starve the OpenMP host maths: keep BLAS single-threaded
    from threadpoolctl import threadpool_limits as _tpl
    _blas_limit = _tpl(limits=1, user_api="blas")
except Exception:  # pragma: no cover
    _blas_limit = None

from .priors import niw_hyperparams, multinomial_hyper, mv_gaussian, multinomial_dist  # noqa: E402,F401
from .api import fit, dp_parallel, predict, run_model_from_checkpoint, resume_from_checkpoint, generate_gaussian_data, generate_mnmm_data, gaussian_mixture_shard, get_labels_histogram, dp_parallel_sampling  # noqa: E402,F401
from .sampler import DPMMSampler, LocalComm  # noqa: E402,F401
from .checkpoint import load_data, save_model, load_checkpoint  # noqa: E402,F401
