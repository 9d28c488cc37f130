"""Race hunt: the same seeded chain, run again and again with random host-side delays in front of every worker call (the native
engine calls libdpmmhip.so through its worker table: the entries are wrapped by Python callbacks that sleep first).  A chain is a pure
function of (data, seed): any run whose K history or labels differ from the first one exposes a host/GPU ordering bug.

    python scripts/race_hunt.py [runs] [max_delay_us] [N] [D] [iters] [device_master -1/0/1]"""
import ctypes
import importlib
import os
import random
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
sampler_mod = importlib.import_module("dpmmsubclusters_jl_amd.host.sampler")

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
delay = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
D = int(sys.argv[4]) if len(sys.argv) > 4 else 64
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 40
dev = int(sys.argv[6]) if len(sys.argv) > 6 else 1
prior = sys.argv[7] if len(sys.argv) > 7 else "niw"

_real_table = engine.native_worker_table
_rng = random.Random(1)
_cur_delay = [0.0]


def jittered_table(worker, rank=0, world=1):
    t, keep = _real_table(worker, rank, world)
    wrappers = []
    for field, sym, ftype in engine._NATIVE_MAP:
        real = ctypes.cast(getattr(worker._lib, sym), ftype)
        if field == "last_error":
            continue

        def make(real):
            def f(*a):
                d = _cur_delay[0]
                if d > 0:
                    r = _rng.random()
                    if r < 0.5:
                        time.sleep(_rng.random() * d * 1e-6)
                return real(*a)
            return f
        cb = ftype(make(real))
        wrappers.append(cb)
        setattr(t, field, cb)
    return t, keep + wrappers


if prior == "niw":
    x, y, _, _ = host.generate_gaussian_data(N, D, 6, 100.0, seed=4242)
    hyper = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
else:
    x, y, _ = host.generate_mnmm_data(N, D, 6, 100, seed=4242)[:3]
    hyper = host.multinomial_hyper(np.ones(D, np.float32))
X = np.ascontiguousarray(np.asarray(x, np.float32).T)
print("host", socket.gethostname(), "cpus", os.cpu_count(), flush=True)


def chain(d):
    _cur_delay[0] = d
    for mod in (engine, sampler_mod):
        if hasattr(mod, "native_worker_table"):
            setattr(mod, "native_worker_table", jittered_table if d >= 0 else _real_table)
    wk = pkg.Worker(hyper.kind, D, N, device=0, seed=99)
    wk.upload_points(X)
    s = host.DPMMSampler(wk, hyper, 10.0, N, 99, burnout=6)
    s.model.set_option(engine.OPT_DEVICE_MASTER, dev)
    s.init_first_clusters(1)
    _, nmi, _, kh = s.run_model(iters, gt=y)
    lab, sub = wk.get_labels()
    wk.close()
    return list(kh), lab, sub


ref = chain(-1)
print("reference K history", ref[0], flush=True)
bad = 0
for r in range(runs):
    d = delay if r % 2 == 0 else delay * 10
    kh, lab, sub = chain(d)
    same = kh == ref[0] and np.array_equal(lab, ref[1]) and np.array_equal(sub, ref[2])
    if not same:
        bad += 1
        print(f"run {r} (delay <= {d} us): DIFFERS  K {kh}  label diffs {(lab != ref[1]).sum() if len(lab) == len(ref[1]) else -1}", flush=True)
print(f"{runs} runs, {bad} differ from the reference chain")
sys.exit(1 if bad else 0)
