"""Growth-trajectory runs (init_clusters=1): BASELINE config 1 (docs example) and config 2 (NIW D=64 N=1e6 32 comps)."""
import sys, time, importlib, json
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
out = {}
# C1: docs/src/getting_started.md:27-37
x, y, _, _ = host.generate_gaussian_data(10 ** 4, 2, 6, 100.0, seed=4)
host.fit(x, 10.0, iters=5, burnout=10, seed=1, verbose=False)          # warm (library load, allocations)
t0 = time.time(); res = host.fit(x, 10.0, iters=100, burnout=10, gt=y, seed=12345, verbose=False); t1 = time.time()
out["C1"] = dict(N=10 ** 4, D=2, iters=100, total_iter_s=float(sum(res[3])), wall_s=t1 - t0, it_per_s=100 / sum(res[3]),
                 K_final=len(res[1]), nmi_final=res[4][-1], reference_docs_it_per_s=93.6)
# C2 growth
N, D, K = 10 ** 6, 64, 32
X, yy = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
t0 = time.time(); res = host.fit(X.T, 10.0, iters=100, burnout=20, gt=yy, seed=123456789, verbose=False); t1 = time.time()
it = np.array(res[3])
out["C2_growth"] = dict(N=N, D=D, iters=100, total_iter_s=float(it.sum()), it_per_s_whole_run=float(100 / it.sum()),
                        it_per_s_last20_nonfinal=float(1 / it[-26:-6].mean()), K_history=res[6][::5], K_final=len(res[1]),
                        nmi_final=res[4][-1], wall_s=t1 - t0)
print(json.dumps(out))
