import sys, os, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
from dpmmsubclusters_jl_amd import binding
D, N, K = 64, 200000, 6
x, y, _, _ = host.generate_gaussian_data(N, D, K, 100.0, seed=4242)
x = np.ascontiguousarray(x, np.float32)
hyper = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
for dev, opts in ((1, {}), (0, {}), (1, {binding.OPT_STATS_DERIVE: 0}), (1, {binding.OPT_BALL_SCREEN: 0})):
    wk = pkg.Worker(hyper.kind, D, N, device=0, seed=99)
    for k, v in opts.items():
        wk.set_option(k, v)
    wk.upload_points(np.ascontiguousarray(x.T))
    s = host.DPMMSampler(wk, hyper, 10.0, N, 99, burnout=8)
    s.model.set_option(engine.OPT_DEVICE_MASTER, dev)
    s.init_first_clusters(1)
    _, nmi, _, kh = s.run_model(60, gt=y)
    print("dev", dev, opts, "K", kh, "nmi", nmi[-1], flush=True)
    wk.close()
