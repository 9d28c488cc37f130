"""Dev helper: wall-clock timeline of individual calls inside group_step at the 8-GPU shard size."""
import sys, time, importlib, functools
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1250000
D, K = 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(22):
    s.group_step(False, False)
log = []
def wrap(obj, name):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); log.append((name, 1e3 * (time.perf_counter() - t0))); return r
    setattr(obj, name, g)
for nm in ("sample_clusters", "update_suff_stats_posterior", "reset_bad_clusters", "check_and_split", "check_and_merge", "remove_empty_clusters"):
    wrap(s, nm)
for nm in ("set_params_niw_chol", "sweep", "suffstats_packed", "reset_sublabels", "split", "merge", "remove_empty", "set_num_clusters", "sync"):
    wrap(wk, nm)
wrap(prior, "sample"); wrap(prior, "update_from_packed"); wrap(prior, "log_marginal"); wrap(prior, "log_marginal_pairs")
for it in range(4):
    log.clear()
    t0 = time.perf_counter(); s.group_step(False, False); tot = 1e3 * (time.perf_counter() - t0)
    print(f"step {it}: {tot:.2f} ms | " + " ".join(f"{n}={t:.2f}" for n, t in log))
print("kernel ms (sweep, stats):", wk.last_kernel_ms(), "threads", host.native.default_threads())
