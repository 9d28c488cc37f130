"""Multinomial chains with the device master (growth from one cluster through splits, merges and removals): K history and a hash of the labels,
sub-labels and packed rows at the end.  Printed, so that two builds can be compared (the chain is a function of the seed alone).
   python scripts/mult_chain_hash.py [steps]"""
import hashlib, importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
engine = importlib.import_module("dpmmsubclusters_jl_amd.host.engine")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
out = []
for D, N, Kt, cnt, seed in [(200, 20000, 5, 150, 3), (160, 8000, 4, 120, 9), (300, 30000, 12, 60, 4)]:
    x, y, _ = host.generate_mnmm_data(N, D, Kt, cnt, seed=seed)[:3]
    x = np.ascontiguousarray(x, np.float32)
    hyper = host.multinomial_hyper(np.ones(D, np.float32))
    wk = pkg.Worker(hyper.kind, D, N, device=0, seed=5)
    wk.upload_points(np.ascontiguousarray(x.T))
    s = host.DPMMSampler(wk, hyper, 10.0, N, 5, burnout=5)
    s.model.set_option(engine.OPT_DEVICE_MASTER, 1)
    s.init_first_clusters(1)
    ks = [s.K]
    for _ in range(steps):
        s.group_step(False, False)
        ks.append(s.K)
    lab, sub = wk.get_labels()
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(lab).tobytes()); h.update(np.ascontiguousarray(sub).tobytes())
    h.update(np.ascontiguousarray(s.model.get("packed")).tobytes())
    out.append({"D": D, "N": N, "K_history": ks, "sha256": h.hexdigest()[:16]})
    wk.close()
print(json.dumps(out))
