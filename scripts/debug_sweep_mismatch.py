"""Dev helper: which points differ between the sweep's labels and a draw from the GPU's own full table?"""
import sys, os, importlib
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from __graft_entry__ import load_package
pkg = load_package()
if os.environ.get("ALT_LIB"):
    b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
    alt = os.path.abspath(os.environ["ALT_LIB"]); b.lib_path = lambda: alt
import test_gpu_niw as T
from oracle import oracle as orc
for case in sys.argv[1:]:
    D, n, K = (int(v) for v in case.split(','))
    P = T.make_problem(D, n, K, seed=11 + D, sep=1.2, sorted_points=False)
    seed, epoch, first = 123456789, 5, 1000003
    wk = T.gpu_worker(pkg, P, seed=seed, first_index=first)
    wk.sweep(epoch)
    lab, sub = wk.get_labels()
    tab = wk.debug_loglik()
    u0, u1 = orc.uniforms(seed, epoch, 0, first, n)
    ref = orc.sample_log_cat(tab, u0)
    bad = np.flatnonzero(ref != lab)
    print("mismatches", len(bad), "of", n, "first", bad[:20])
    for i in bad[:6]:
        col = tab[:, i]
        print(i, "tile", i // 64, "lane", i % 64, "gpu", lab[i], "ref", ref[i], "u", u0[i], "a-max:", np.round(col - col.max(), 2))
    print("tiles with mismatches:", np.unique(bad // 64)[:20], "lanes:", np.unique(bad % 64)[:64])
    wk.close()
