"""Dev helper: the growth run of bench.py (N=1e7, D=64, from ONE cluster) with the direction screen in automatic mode (or argv[1] = 0 / 1):
per iteration the step time, K and the sweep's work counters per wave tile -- when does the screen switch on, what does it remove."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
mode = float(sys.argv[1]) if len(sys.argv) > 1 else -1.0
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 7
D, K = 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
wk.set_option(23, mode)
g = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
g.init_first_clusters(1)
rows = []
t_last = [time.perf_counter()]
def cb(i, s):
    w = wk.last_sweep_work()
    t = max(1.0, w["wave_tiles"])
    now = time.perf_counter()
    rows.append((i, s.K, 1e3 * (now - t_last[0]), w["direction_screens"] / t, w["bf16_bottom_screens"] / t, w["bf16_top_screens"] / t, w["screens16"] / t, w["full_evals"] / t, w["tail_pairs"] / t))
    t_last[0] = time.perf_counter()
g.run_model(int(sys.argv[3]) if len(sys.argv) > 3 else 160, gt=None, on_iteration=cb)
print("iter  K   ms(incl. counter read)  direction  bf16-bottom  bf16-top  f32-screens  full-evals  tail-pairs   (per wave tile)")
for r in rows:
    if r[0] >= (int(sys.argv[4]) if len(sys.argv) > 4 else 20):
        print("%4d %3d %8.3f   %6.2f %8.2f %8.2f %8.2f %8.2f %8.2f" % r)
