"""Dev helper: A/B of a dpmm_set_option switch on the headline workload, interleaved rounds in ONE process.
   python3 scripts/ab_option.py <option id> <value A> <value B> [points] [rounds] [id=value ...]     (further options, set once for both legs)"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
opt, va, vb = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
N = int(float(sys.argv[4])) if len(sys.argv) > 4 else 10 ** 7
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 6
D, K = 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(10):
    s.group_step(False, False)
for kv in sys.argv[6:]:
    wk.set_option(int(kv.split("=")[0]), float(kv.split("=")[1]))
res = {va: [], vb: []}
st = {va: [], vb: []}
wall = {va: [], vb: []}
for r in range(rounds):
    for v in (va, vb):
        wk.set_option(opt, v)
        s.group_step(False, False); s.group_step(False, False)
        ms, ss = [], []
        for _ in range(5):
            s.group_step(False, False); a, b = wk.last_kernel_ms(); ms.append(a); ss.append(b)
        res[v].append(float(np.median(ms))); st[v].append(float(np.median(ss)))
        wk.set_timing(False); wk.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            s.group_step(False, False)
        wk.sync(); wall[v].append((time.perf_counter() - t0) / 20 * 1e3)
        wk.set_timing(True)
for v in (va, vb):
    print(f"option {opt} = {v}: sweep kernel median {np.median(res[v]):.4f} ms  min {np.min(res[v]):.4f}  rounds {np.round(res[v], 4).tolist()}; "
          f"statistics pass median {np.median(st[v]):.4f} ms; step without timing events median {np.median(wall[v]):.4f} ms  rounds {np.round(wall[v], 4).tolist()}")
