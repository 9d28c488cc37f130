"""Dev helper: the draw-ahead / normals-ahead machinery against the draw-when-asked chain over many seeds and shapes (bitwise), and
repeated runs of one configuration against each other (run-to-run determinism).  python3 scripts/stress_ahead.py [rounds]"""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def run(D, N, Kt, seed, ahead, iters=40):
    X, y = host.gaussian_mixture_shard(N, D, Kt, 100.0, 1000 + seed, 0, N)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=seed)
    wk.upload_points(X)
    s = host.DPMMSampler(wk, prior, 10.0, N, seed, burnout=3)
    s._configure()
    s.model.set_option(engine.OPT_DEVICE_MASTER, 1)
    s.model.set_option(engine.OPT_DRAW_AHEAD, ahead)
    s.init_first_clusters(1)
    tr = []
    for it in range(iters):
        s.group_step(it >= iters - 4, False)
        tr.append(s.K)
    lab, sub = wk.get_labels()
    p = s.params
    out = (tr, lab.copy(), sub.copy(), p["mu"].copy(), p["R"].copy(), s.model.get("log_marginal").copy())
    wk.close()
    return out


def same(a, b):
    return a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))


bad = 0
for r in range(rounds):
    for D, N, Kt in ((16, 20000, 5), (70, 20000, 4), (130, 15000, 3), (256, 12000, 2)):
        a = run(D, N, Kt, 11 + r, 1)
        b = run(D, N, Kt, 11 + r, 0)
        c = run(D, N, Kt, 11 + r, 1)
        ok = same(a, b) and same(a, c)
        bad += not ok
        print(f"round {r} D={D}: K history end {a[0][-1]} max {max(a[0])}  ahead == asked: {same(a, b)}  run-to-run: {same(a, c)}", flush=True)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
