"""Dev / evidence helper: steady-state `group_step` at one of the BASELINE shapes on ONE GPU, through the native engine.
   python3 scripts/config_step.py niw 256 625000 [steps]       C5's per-GPU shard (NIW D=256, n=6.25e5, K=32)
   python3 scripts/config_step.py niw 64 1250000               C3's per-GPU shard
   python3 scripts/config_step.py mult 1000 1000000            C4 (Multinomial D=1000, N=1e6, K=32)
Prints one JSON line: ms per step, kernel times (HIP events), host timers per step, algorithmic bytes / flops rates.
Run under `rocprofv3 --kernel-trace --stats` / `--pmc ...` for the tracked profiles (scripts/collect_profiles.sh)."""
import importlib
import json, os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
kind = sys.argv[1] if len(sys.argv) > 1 else "niw"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(float(sys.argv[3])) if len(sys.argv) > 3 else 625000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
timing = not (len(sys.argv) > 5 and sys.argv[5] == "notiming")      # "notiming": no HIP events between the kernels (as fit / dp_parallel run)
rccl1 = len(sys.argv) > 6 and sys.argv[6] == "rccl1"                  # attach a ONE-rank RCCL communicator: the library's collective path, no wire
mixvar = float(sys.argv[7]) if len(sys.argv) > 7 else 100.0           # variance of the component means (4 / 1: overlapping clusters, the direction screen's regime)
K, burnout = 32, 20
if kind == "niw":
    X, y = host.gaussian_mixture_shard(N, D, K, mixvar, 12345, 0, N)
    prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
    wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
else:
    rng = np.random.default_rng(0)
    P = rng.dirichlet(np.ones(D) * 0.5, size=K)
    y = rng.integers(1, K + 1, N)
    X = np.empty((N, D), np.float32)
    for k in range(K):
        m = y == k + 1
        X[m] = rng.multinomial(100, P[k], size=int(m.sum()))
    prior = host.multinomial_hyper(np.ones(D, np.float32))      # test/save_load_test/multinomial_params.jl:24
    wk = pkg.Worker(pkg.PRIOR_MULT, D, N, device=0, seed=1)
for kv in os.environ.get("DPMM_STEP_OPTS", "").split(","):           # e.g. DPMM_STEP_OPTS=12=256,4=0 : library options (include/dpmm_hip.h) before the upload
    if kv:
        wk.set_option(int(kv.split("=")[0]), float(kv.split("=")[1]))
wk.upload_points(X)
if rccl1:
    wk.comm_init(wk.comm_unique_id(), 0, 1)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=burnout)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for _ in range(burnout + 5):
    s.group_step(False, False)
wk.set_timing(timing)
t_before = dict(s.timers)
ts, sw, st = [], [], []
for _ in range(steps):
    t0 = time.perf_counter(); s.group_step(False, False); ts.append(time.perf_counter() - t0)
    a, b = wk.last_kernel_ms() if timing else (float("nan"), float("nan")); sw.append(a); st.append(b)
t_after = dict(s.timers)
lab, _ = wk.get_labels()
out = {"config": f"{kind} D={D} N={N} K={s.K}", "ms_per_step": 1e3 * float(np.mean(ts)), "ms_per_step_min": 1e3 * float(np.min(ts)),
       "sweep_kernel_ms": float(np.mean(sw)), "stats_kernels_ms": float(np.mean(st)), "label_agreement": float(np.mean(lab == y)),
       "splittable": int(s.splittable.sum()), "bad_resets_total_and_steps": [int(v) for v in s.model.get("counters")[4:6]],
       "min_sub_occupancy": float(np.min(s.N[:, 1:3])), "sub_occupancy_sorted_head": np.sort(np.min(s.N[:, 1:3], axis=1))[:6].tolist(),
       "host_ms_per_step": {k: round(1e3 * (t_after[k] - t_before[k]) / steps, 4) for k in t_after}}
if kind == "niw":
    out["algorithmic_tflops_sweep"] = 2.0 * N * D * D * (s.K + 2) / (np.mean(sw) * 1e-3) / 1e12
    w = wk.last_sweep_work()
    out["executed_tflops_sweep"] = w["executed_flops"] / (sw[-1] * 1e-3) / 1e12
    out["work"] = w
else:
    out["algorithmic_GBps_sweep"] = (4.0 * N * D + 16.0 * N) / (np.mean(sw) * 1e-3) / 1e9
    out["algorithmic_GBps_sweep_plus_stats"] = (4.0 * N * D + 16.0 * N) / ((np.mean(sw) + np.mean(st)) * 1e-3) / 1e9
print(json.dumps(out))
wk.close()
