import sys, os, importlib, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
print("host", socket.gethostname(), "cpus", os.cpu_count(), "aff", len(os.sched_getaffinity(0)), torch.cuda.get_device_properties(0).multi_processor_count, flush=True)
D, N, K = 64, 200000, 6
x, y, _, _ = host.generate_gaussian_data(N, D, K, 100.0, seed=4242)
x = np.ascontiguousarray(x, np.float32)
hyper = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
def poison(pat):
    if pat is None:
        return
    free, total = torch.cuda.mem_get_info()
    n = int(min(free * 0.5, 40e9))
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    if pat == "rand":
        t.random_(0, 256)
    else:
        t.fill_(pat)
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()
for pat in (None, 0xFF, 0x7F, "rand", 0x01, 0x80):
    poison(pat)
    wk = pkg.Worker(hyper.kind, D, N, device=0, seed=99)
    wk.upload_points(np.ascontiguousarray(x.T))
    s = host.DPMMSampler(wk, hyper, 10.0, N, 99, burnout=8)
    s.model.set_option(engine.OPT_DEVICE_MASTER, 1)
    s.init_first_clusters(1)
    _, nmi, _, kh = s.run_model(60, gt=y)
    print("poison", pat, "K", kh[-1], "at20", kh[20], "nmi", round(float(nmi[-1]), 4), flush=True)
    wk.close()
