"""Randomised check of the derived per-step statistics (DPMM_OPT_STATS_DERIVE): random sequences of sweeps with random parameters,
partial relabelling, split / merge / remove-empty with K changing, option toggles and re-uploads; after EVERY per-step pass the rows
must equal the oracle's from-scratch statistics of the labels left behind (rtol 1e-12) and N exactly.
   python3 scripts/fuzz_derive.py [rounds] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from __graft_entry__ import load_package
from oracle import oracle as orc
from test_gpu_niw import make_problem
pkg = load_package()
from dpmmsubclusters_jl_amd import binding
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
checks = 0
for trial in range(6):
    D = int(rng.choice([8, 16, 40, 64, 130]))
    n = int(rng.integers(3000, 20000))
    K = int(rng.integers(2, 9))
    P = make_problem(D, n, K, seed=int(rng.integers(1 << 30)), sep=float(rng.uniform(0.8, 3.0)))
    X = P["X"]
    wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=int(rng.integers(1 << 30)))
    wk.upload_points(X)
    lab = rng.integers(1, K + 1, n); sub = rng.integers(1, 3, n)
    wk.set_labels(lab, sub); wk.set_num_clusters(K)
    epoch = 1
    for it in range(rounds // 6):
        op = rng.choice(["sweep", "sweep", "sweep", "sub", "move", "split", "merge", "remove", "toggle", "upload", "none"])
        if op == "sweep":
            Kc = wk.K
            Q = make_problem(D, 10, Kc, seed=int(rng.integers(1 << 30)), sep=float(rng.uniform(0.5, 3.0)))
            wk.set_params_niw(Q["mu"], Q["invS"], Q["logdet"], Q["lr"], Q["w"]); epoch += 1; wk.sweep(epoch)
        elif op == "sub":
            l, s = wk.get_labels(); m = rng.random(n) < rng.uniform(0, 0.6); s[m] = 3 - s[m]; wk.set_labels(l, s)
        elif op == "move":
            l, s = wk.get_labels(); m = rng.random(n) < rng.choice([0.0002, 0.01, 0.3]); l[m] = rng.integers(1, wk.K + 1, int(m.sum())); wk.set_labels(l, s)
        elif op == "split" and wk.K < 12:
            Kc = wk.K; k = int(rng.integers(1, Kc + 1)); wk.set_num_clusters(Kc + 1); epoch += 1; wk.split(np.array([k]), np.array([Kc + 1]), epoch)
        elif op == "merge" and wk.K >= 2:
            i, j = sorted(rng.choice(np.arange(1, wk.K + 1), 2, replace=False)); wk.merge(np.array([i]), np.array([j]))
        elif op == "remove":
            l, _ = wk.get_labels(); cnt = np.bincount(l, minlength=wk.K + 1)[1:wk.K + 1]
            if (cnt == 0).any() and (cnt > 0).any():
                wk.remove_empty(cnt); wk.set_num_clusters(int((cnt > 0).sum()))
        elif op == "toggle":
            wk.set_option(binding.OPT_STATS_DERIVE, int(rng.integers(0, 2)))
        elif op == "upload":
            X = (X + np.float32(rng.normal() * 0.1)).astype(np.float32); wk.upload_points(X)
        epoch += 1
        packed, bad = wk.step_stats(epoch)
        l, s = wk.get_labels(); Kc = wk.K
        N, sm, S = wk.unpack(packed, Kc)
        oN, os_, oS = orc.suffstats_niw(X, D, l, s, Kc)
        assert np.array_equal(N, oN), (trial, it, op)
        np.testing.assert_allclose(sm, os_, rtol=1e-12, atol=1e-9, err_msg=f"{trial} {it} {op}")
        np.testing.assert_allclose(S, oS, rtol=1e-12, atol=1e-8, err_msg=f"{trial} {it} {op}")
        checks += 1
    wk.set_option(binding.OPT_STATS_DERIVE, 1)
    wk.close()
print(f"fuzz_derive: {checks} per-step passes checked against the oracle, all equal")
