"""End-to-end sanity + timing on the other BASELINE config shapes (reduced N): NIW D=256 and Multinomial D=1000."""
import sys, time, importlib, json
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
out = {}
X, y = host.gaussian_mixture_shard(200000, 256, 8, 100.0, 5, 0, 200000)
t0 = time.time(); res = host.fit(X.T, 10.0, iters=60, burnout=8, gt=y, seed=3, verbose=False); t1 = time.time()
it = np.array(res[3])
out["niw_d256"] = dict(N=200000, K_true=8, K_final=len(res[1]), nmi=res[4][-1], ms_per_iter_last10=float(1e3 * it[-10:].mean()), wall_s=t1 - t0, K_hist=res[6][::6])
x, lab, _ = host.generate_mnmm_data(200000, 1000, 16, 100, seed=1)
hyper = host.multinomial_hyper(np.ones(1000, np.float32))
t0 = time.time(); res = host.fit(x, hyper, 10.0, iters=60, burnout=8, gt=lab, seed=3, verbose=False); t1 = time.time()
it = np.array(res[3])
out["mult_d1000"] = dict(N=200000, K_true=16, K_final=len(res[1]), nmi=res[4][-1], ms_per_iter_last10=float(1e3 * it[-10:].mean()), wall_s=t1 - t0, K_hist=res[6][::6])
print(json.dumps(out))
