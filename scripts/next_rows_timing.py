"""Timing of the SURVEY 8(f) "next row" device paths at bench scale (one GPU): predict, contingency, .npy ingestion, smart splits."""
import sys, time, importlib, json
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
out = {}
N, D, K = 10 ** 7, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
t0 = time.perf_counter(); wk.upload_points(X); t1 = time.perf_counter()
out["upload_points_f32_GBps"] = X.nbytes / (t1 - t0) / 1e9
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(3):
    s.group_step(False, False)
def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts)
# on-device evaluation
wk.set_ground_truth_range(y - y.min(), int(y.max() - y.min() + 1))
out["contingency_ms_N1e7"] = best(lambda: wk.contingency(s.K))
out["gather_labels_ms_N1e7 (what the reference does per iteration)"] = best(lambda: wk.get_labels(), 3)
# smart splits, worker halves on the largest cluster
k = int(np.argmax(s.N[:, 0])); mu = s.sums[k, 0] / s.N[k, 0]
v = np.zeros(D); v[0] = 1.0
out["smart_cluster_points"] = int(s.N[k, 0])
out["smart_project_ms"] = best(lambda: wk.smart_project(k + 1, v, mu), 3)
out["smart_kmeans_iter_ms"] = best(lambda: wk.smart_kmeans_iter(k + 1, -1.0, 1.0))
_, sub0 = wk.get_labels()
out["smart_assign_ms"] = best(lambda: (wk.smart_assign(k + 1, -1.0, 1.0), wk.sync()))
wk.set_labels(None, sub0)
wk.close()
# predict: 1e6 new points against the fitted model
class M: pass
m = M(); m.sampler = s
Xp = np.ascontiguousarray(X[:10 ** 6].T)
host.predict(m, Xp)
out["predict_ms_1e6_points_K32 (context + upload + table + device finish + download)"] = best(lambda: host.predict(m, Xp), 3)
# .npy ingestion on the device: Float64 rows with NaNs
rows = X[:2 * 10 ** 6].astype(np.float64); rows[::1000, 3] = np.nan
w2 = pkg.Worker(pkg.PRIOR_NIW, D, len(rows), device=0, seed=1)
w2.upload_points_npy(rows)
t = best(lambda: w2.upload_points_npy(rows), 3)
out["upload_points_npy_f64_GBps (host bytes / time, pageable source)"] = rows.nbytes / (t * 1e-3) / 1e9
w2.close()
print(json.dumps(out))
