"""Dev helper: single- and multi-thread timings of the native host maths (posterior, noise, parameter draws, merge pairs) at K=32, D=64."""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
native = importlib.import_module("dpmmsubclusters_jl_amd.host.native")
D, K = 64, 32
rng = np.random.default_rng(0)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
n = 3 * K
N = rng.integers(1000, 300000, n).astype(float)
X = [rng.normal(size=(200, D)) for _ in range(n)]
sums = np.stack([x.sum(0) * (N[i] / 200) for i, x in enumerate(X)])
S = np.stack([x.T @ x * (N[i] / 200) for i, x in enumerate(X)])
def bench(f, reps=20):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return 1e3 * (time.perf_counter() - t0) / reps
for nt in (1, 14):
    post = prior.posterior(N, sums, S, nthreads=nt)
    print("threads", nt)
    print("  posterior      %.3f ms" % bench(lambda: prior.posterior(N, sums, S, nthreads=nt)))
    noise = prior.draw_noise(n, 1, 5, nthreads=nt)
    print("  draw_noise     %.3f ms" % bench(lambda: prior.draw_noise(n, 1, 5, nthreads=nt)))
    print("  sample(noise)  %.3f ms" % bench(lambda: prior.sample(post, 1, 5, np.arange(n), nthreads=nt, noise=noise)))
    print("  log_marginal   %.3f ms" % bench(lambda: prior.log_marginal(post, N)))
    ii, jj = np.triu_indices(K, 1)
    pairs = np.stack([3 * ii, 3 * jj], 1)
    print("  pairs(496)     %.3f ms" % bench(lambda: prior.log_marginal_pairs(pairs, dict(N=N, sums=sums, S=S), nthreads=nt)))
