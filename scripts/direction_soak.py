"""Dev helper: a long chain on overlapping components (MixtureVar 2, N = 1e6, D = 64, from the generator's labels) with the direction screen in
automatic mode and with it off: step times, how often the screen ran, and whether the two chains are the same chain after `steps` steps."""
import importlib, json, sys, time
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N, D, K = 10 ** 6, 64, 24
X, y = host.gaussian_mixture_shard(N, D, K, 2.0, 4242, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
out = {}
for mode in (-1, 0):
    wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=77)
    wk.upload_points(X)
    wk.set_option(23, mode)
    s = host.DPMMSampler(wk, prior, 10.0, N, 77, burnout=10)
    s.start_from_labels(y, 1 + np.random.default_rng(3).integers(0, 2, N), K)
    wk.last_sweep_work()
    t0 = time.perf_counter()
    ks = []
    for it in range(steps):
        s.group_step(False, False)
        ks.append(s.K)
    wk.sync()
    el = time.perf_counter() - t0
    w = wk.last_sweep_work()
    lab, sub = wk.get_labels()
    out[mode] = dict(ms_per_step=1e3 * el / steps, K_final=int(s.K), K_changes=int(np.count_nonzero(np.diff(ks))), direction_screens_per_tile=w["direction_screens"] / max(1.0, w["wave_tiles"]),
                     bf16_bottom_per_tile=w["bf16_bottom_screens"] / max(1.0, w["wave_tiles"]))
    out[mode]["_lab"] = lab; out[mode]["_sub"] = sub; out[mode]["_ks"] = ks
    wk.close()
same = bool(np.array_equal(out[-1]["_lab"], out[0]["_lab"]) and np.array_equal(out[-1]["_sub"], out[0]["_sub"]) and out[-1]["_ks"] == out[0]["_ks"])
for m in out:
    for k in ("_lab", "_sub", "_ks"):
        out[m].pop(k)
print(json.dumps({"steps": steps, "automatic": out[-1], "off": out[0], "same_chain": same}))
