"""Dev helper: random (D, K, separation, conditioning, seed) problems -- the labels of the D <= 64 sweep with EVERY screen on (tail, ball, bracket,
bf16, direction screen forced on) against the same sweep with screening off (margin 0: every cluster evaluated in full), two epochs each.
   python3 scripts/fuzz_screens.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    D = int(rng.choice([36, 40, 48, 52, 56, 60, 64]))
    K = int(rng.integers(2, 41))
    sep = float(np.exp(rng.uniform(np.log(0.3), np.log(30.0))))
    cond = float(np.exp(rng.uniform(0.0, np.log(300.0))))          # spread of the covariance spectrum
    n = 20000
    mus = rng.normal(size=(3 * K, D)) * sep
    for k in range(K):
        d = rng.normal(size=D) * 0.4
        mus[3 * k + 1] = mus[3 * k] + d; mus[3 * k + 2] = mus[3 * k] - d
    Sig = np.empty((3 * K, D, D))
    for j in range(3 * K):
        Q, _ = np.linalg.qr(rng.normal(size=(D, D)))
        ev = np.exp(rng.uniform(-0.5 * np.log(cond), 0.5 * np.log(cond), D))
        Sig[j] = (Q * ev) @ Q.T
    invS = np.linalg.inv(Sig); invS = 0.5 * (invS + invS.transpose(0, 2, 1))
    logdet = np.linalg.slogdet(Sig)[1]
    z = np.sort(rng.integers(0, K, n))
    L = np.linalg.cholesky(Sig[3 * z])
    X = (mus[3 * z] + np.einsum("nij,nj->ni", L, rng.normal(size=(n, D)))).astype(np.float32)
    w = rng.dirichlet(np.ones(K) * 5).astype(np.float32); lr = rng.dirichlet(np.ones(2) * 5, size=K).astype(np.float32)
    labs = {}
    for mode in ("screens", "dense"):
        wk = pkg.Worker(pkg.PRIOR_NIW, D, n, device=0, seed=1000 + case)
        wk.upload_points(X)
        if mode == "dense":
            wk.set_option(1, 0.0)            # DPMM_OPT_SCREEN_MARGIN = 0
        else:
            wk.set_option(23, 1.0)           # direction screen forced on
        wk.set_labels(z + 1, 1 + (np.arange(n) & 1))
        wk.set_params_niw(mus.astype(np.float32), invS.reshape(3 * K, -1).astype(np.float32), logdet.astype(np.float32), lr, w)
        wk.suffstats_packed(None)                       # the bin-sorted visiting order
        out = []
        for ep in (1, 2):
            wk.set_params_niw(mus.astype(np.float32), invS.reshape(3 * K, -1).astype(np.float32), logdet.astype(np.float32), lr, w)
            wk.sweep(ep)
            out.append(wk.get_labels())
            wk.suffstats_packed(None)
        labs[mode] = out
        if mode == "screens":
            work = wk.last_sweep_work()
        wk.close()
    ok = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(labs["screens"], labs["dense"]))
    t = max(1.0, work["wave_tiles"])
    print(f"case {case:3d} D={D} K={K:2d} sep={sep:6.2f} cond={cond:6.1f}: {'same' if ok else 'DIFFERENT'}   full evals/tile {work['full_evals'] / t:5.2f} of {K + 2}, direction screens/tile {work['direction_screens'] / t:4.2f}")
    bad += not ok
print("cases", cases, "different", bad)
sys.exit(1 if bad else 0)
