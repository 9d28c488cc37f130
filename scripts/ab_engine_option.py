"""Dev helper: A/B of a dpmmh_model_set_option switch (engine side), interleaved rounds in ONE process: step time + host timers.
   python3 scripts/ab_engine_option.py <option id> <value A> <value B> [points] [rounds] [D]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
opt, va, vb = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
N = int(float(sys.argv[4])) if len(sys.argv) > 4 else 1250000
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 6
D = int(sys.argv[6]) if len(sys.argv) > 6 else 64
K = 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(25):
    s.group_step(False, False)
res = {va: [], vb: []}
tim = {va: {}, vb: {}}
for r in range(rounds):
    for v in (va, vb):
        s.model.set_option(opt, v)
        for _ in range(3):
            s.group_step(False, False)
        t_before = dict(s.timers)
        t0 = time.perf_counter()
        for _ in range(20):
            s.group_step(False, False)
        res[v].append((time.perf_counter() - t0) / 20 * 1e3)
        t_after = dict(s.timers)
        for k in t_after:
            tim[v].setdefault(k, []).append(1e3 * (t_after[k] - t_before[k]) / 20)
for v in (va, vb):
    print(f"option {opt} = {v}: step median {np.median(res[v]):.4f} ms  min {np.min(res[v]):.4f}  rounds {np.round(res[v], 4).tolist()}")
    print("   ", {k: round(float(np.median(x)), 4) for k, x in tim[v].items() if np.median(x) > 0.001})
