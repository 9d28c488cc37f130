"""Dev helper: A/B of a dpmm_set_option switch on another shape (NIW, D and N given), interleaved rounds in ONE process.
   python3 scripts/ab_option_cfg.py <option id> <value A> <value B> D N [rounds]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
opt, va, vb = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
D, N = int(sys.argv[4]), int(float(sys.argv[5]))
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 5
K = 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(25):
    s.group_step(False, False)
res = {va: [], vb: []}; wall = {va: [], vb: []}
for r in range(rounds):
    for v in (va, vb):
        wk.set_option(opt, v)
        s.group_step(False, False); s.group_step(False, False)
        ms = []
        for _ in range(5):
            s.group_step(False, False); ms.append(wk.last_kernel_ms()[0])
        res[v].append(float(np.median(ms)))
        wk.set_timing(False); wk.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            s.group_step(False, False)
        wk.sync(); wall[v].append((time.perf_counter() - t0) / 20 * 1e3)
        wk.set_timing(True)
lab, _ = wk.get_labels()
for v in (va, vb):
    print(f"D={D} N={N} option {opt} = {v}: sweep kernel median {np.median(res[v]):.4f} ms  rounds {np.round(res[v], 4).tolist()}; step median {np.median(wall[v]):.4f} ms; labels = generator {np.mean(lab == y):.5f}")
