"""Dev helper: longer runs -- (1) growth from one cluster to all 32 components at N=1e6, D=64 (capacity growth, split/merge churn),
(2) 3000 steady-state sweeps at N=1e6 (time per step must stay flat, device memory must not grow)."""
import sys, time, importlib, json, subprocess
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
out = {}
N, D, K = 10 ** 6, 64, 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
t0 = time.time(); res = host.fit(X.T, 10.0, iters=260, burnout=20, gt=y, seed=123456789, verbose=False); t1 = time.time()
out["growth"] = dict(iters=260, K_history=res[6][::20], K_final=len(res[1]), nmi_final=res[4][-1], wall_s=t1 - t0)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
def vram():
    try:
        import torch
        free, total = torch.cuda.mem_get_info(0)
        return (total - free) / 2 ** 20
    except Exception:
        return -1.0
ts = []
m0 = None
for i in range(3000):
    t0 = time.perf_counter(); s.group_step(False, False); ts.append(time.perf_counter() - t0)
    if i == 100: m0 = vram()
m1 = vram()
ts = np.array(ts) * 1e3
out["soak"] = dict(steps=3000, ms_first500=float(ts[100:600].mean()), ms_last500=float(ts[-500:].mean()), ms_max=float(ts[100:].max()),
                   K_final=int(s.K), vram_mib_after_100=m0, vram_mib_after_3000=m1)
lab, _ = wk.get_labels()
out["soak"]["label_agreement"] = float((lab == y).mean())
print(json.dumps(out))
