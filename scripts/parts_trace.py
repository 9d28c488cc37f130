"""Dev helper: the three launches of every sweep of a steady-state run (N [1e7]) with the sweep's executed-work counters:
   python3 scripts/parts_trace.py [N]"""
import sys, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
D, K = 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=123456789)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 123456789, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(30):
    s.group_step(False, False)
wk.set_timing(15)
rows = []
for i in range(24):
    s.group_step(False, False)
    p = wk.last_sweep_parts_ms(); w = wk.last_sweep_work()
    rows.append((p[0], p[1], p[2], w["full_evals"], w["screens16"], w["bf16_top_screens"], w["bf16_bottom_screens"], w["brackets"], w["tail_pairs"], w["b3_evals"]))
for r in rows: print("lean %.3f  labels %.3f  sub %.3f   full evals %.0f  screens16 %.0f  top %.0f bottom %.0f brackets %.0f tailpairs %.0f b3 %.0f" % r)
