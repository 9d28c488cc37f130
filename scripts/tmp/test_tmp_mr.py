import os, sys
import numpy as np
import pytest
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def child(rank, out, opts, eopts):
    import importlib
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_multirank as t
    from __graft_entry__ import load_package
    pkg = load_package()
    host = importlib.import_module("dpmmsubclusters_jl_amd.host")
    engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
    prior, D, N, K, iters, burnout, dev = t.CASES["niw64_dev"]
    x, y, hyper = t._data(host, "niw64_dev")
    wk = pkg.Worker(hyper.kind, D, N, first_index=0, device=0, seed=99)
    for k, v in opts.items():
        wk.set_option(k, v)
    wk.upload_points(np.ascontiguousarray(x.T))
    s = host.DPMMSampler(wk, hyper, 10.0, N, 99, burnout=burnout)
    s.model.set_option(engine.OPT_DEVICE_MASTER, dev)
    for k, v in eopts.items():
        s.model.set_option(k, v)
    s.init_first_clusters(1)
    _, nmi, _, kh = s.run_model(iters, gt=y)
    print("opts", opts, eopts, "K", kh[-1], "at20", kh[20], "nmi", nmi[-1], flush=True)
    wk.close()

@pytest.mark.gpu
@pytest.mark.parametrize("opts,eopts", [({}, {}), ({16: 0}, {}), ({14: 0}, {}), ({15: 0}, {}), ({}, {9: 0}), ({}, {3: 1}), ({}, {6: 0})])
def test_x(opts, eopts, tmp_path):
    mp.spawn(child, args=(str(tmp_path / "o.npz"), opts, eopts), nprocs=1, join=True)
