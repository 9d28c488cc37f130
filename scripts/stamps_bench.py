"""Dev helper: phase stamps (DPMM_STAMPS build) on the real bench data/parameters."""
import sys, os, ctypes, importlib
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
b = importlib.import_module("dpmmsubclusters_jl_amd.binding")
alt = os.path.abspath("dpmmsubclusters.jl_amd/lib/libdpmmhip_stamps.so")
b.lib_path = lambda: alt
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
K = 32
X, y = host.gaussian_mixture_shard(N, D, K, 100.0, 12345, 0, N)
prior = host.niw_hyperparams(1.0, np.zeros(D), D + 3, np.eye(D))
wk = pkg.Worker(pkg.PRIOR_NIW, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
for kv in sys.argv[3:]:                        # id=value ... (dpmm_set_option)
    wk.set_option(int(kv.split("=")[0]), float(kv.split("=")[1]))
for _ in range(6):
    s.group_step(False, False)
print("kernel ms (stamped build)", wk.last_kernel_ms()[0])
lib = b.load_library()
nw = 4 * 4096
buf = np.zeros((nw, 16), np.uint64)
lib.dpmm_dev_stamps.restype = ctypes.c_int
used = lib.dpmm_dev_stamps(wk._h, buf.ctypes.data_as(ctypes.c_void_p), nw)
d = buf[:used].astype(np.float64); d = d[d[:, 8] > 0]
names = ["x load", "refs(full)", "screen setup", "K-loop", "survivors", "draw", "phase2", "total"] if D <= 64 else ["x load", "reference eval", "-", "tail screens", "survivors", "draw", "phase2 (sub-labels)", "total"]
nt = d[:, 8].sum()
for i, nm in enumerate(names):
    print(f"{nm:14s} cycles/tile {d[:, i].sum() / nt:10.0f}   share {100 * d[:, i].sum() / d[:, 7].sum():5.1f}%")
print("refs setup (table init, prev labels, first fragments, pre-screen norms) cycles/tile:", d[:, 12].sum() / nt)
print("tail-screened clusters per tile:", d[:, 9].sum() / nt)
print("MFMA-screened clusters per tile:", d[:, 10].sum() / nt)
tt = d[:, 7]
print("per-wave total cycles: min %.0f  median %.0f  p90 %.0f  max %.0f ; tiles per wave min %d max %d" % (tt.min(), np.median(tt), np.percentile(tt, 90), tt.max(), d[:, 8].min(), d[:, 8].max()))
wg = tt.reshape(-1, 4).max(axis=1)
order = np.argsort(-wg)[:8]
print("slowest workgroups (index, cycles):", [(int(i), int(wg[i])) for i in order])
for i, nm in enumerate(names[:7]):
    print(f"  {nm:22s} per-wave sum: median {np.median(d[:, i]):10.0f}  max {d[:, i].max():10.0f}")
if D > 64:
    print("longest single tile per wave: median %.0f  max %.0f cycles;  wave lifetime (first stamp -> last stamp): median %.0f max %.0f; spread of end times %.0f" % (
        np.median(d[:, 10]), d[:, 10].max(), np.median(d[:, 12] - d[:, 11]), (d[:, 12] - d[:, 11]).max(), d[:, 12].max() - d[:, 12].min()))
if D <= 64:
    idx = np.argsort(-tt)[:12]
    print("slowest waves: wave | x refs setup Kloop surv draw phase2 total | tiles tail-screened mfma-screened")
    for i in idx:
        print(int(i), [int(v) for v in d[i, :8]], [int(v) for v in d[i, 8:11]])
    med = np.argsort(tt)[len(tt) // 2]
    print("median wave", int(med), [int(v) for v in d[med, :8]], [int(v) for v in d[med, 8:11]])
    t0 = d[:, 14].min()
    print("wave starts (first stamp - earliest): median %.0f max %.0f; ends: median %.0f max %.0f; longest tile: median %.0f p99 %.0f max %.0f" % (
        np.median(d[:, 14] - t0), (d[:, 14] - t0).max(), np.median(d[:, 15] - t0), (d[:, 15] - t0).max(), np.median(d[:, 11]), np.percentile(d[:, 11], 99), d[:, 11].max()))
    big = np.argsort(-d[:, 11])[:40]
    print("longest tiles: (wave, cycles, start offset)", [(int(i), int(d[i, 11]), int(d[i, 13] - t0)) for i in big])
    rel = (d[:, 13] - d[:, 14])
    sel = d[:, 11] > 1.8 * np.median(d[:, 11])
    print("waves with a tile > 1.8 x median longest:", int(sel.sum()), "of", len(d), "; that tile began (cycles after the wave's first stamp) percentiles 0/25/50/75/100:",
          [int(v) for v in np.percentile(rel[sel], [0, 25, 50, 75, 100])], "; wave lifetime median", int(np.median(d[:, 15] - d[:, 14])))
    print("histogram of the long tile's start / 61k:", np.bincount((rel[sel] / 61000).astype(int), minlength=12).tolist())
    print("waves hit, by workgroup index mod 8 (XCD):", np.bincount((np.nonzero(sel)[0] // 4) % 8, minlength=8).tolist())
if D <= 64:
    srt = np.sort(tt)[::-1]
    print("top-20 per-wave totals / median:", np.round(srt[:20] / np.median(tt), 3).tolist())
    print("percentiles 50/90/99/99.9/100 of total/median:", np.round(np.percentile(tt, [50, 90, 99, 99.9, 100]) / np.median(tt), 3).tolist())
    top = np.argsort(-tt)[:12]
    print("slowest waves: (wave, tiles, total/median, longest tile)", [(int(i), int(d[i, 8]), round(float(tt[i] / np.median(tt)), 3), int(d[i, 11])) for i in top])
    for ntile in (int(d[:, 8].min()), int(d[:, 8].max())):
        m = d[:, 8] == ntile
        print(f"waves with {ntile} tiles: {int(m.sum())}, total median {np.median(tt[m]):.0f} max {tt[m].max():.0f}")
    w = int(np.argmax(d[:, 11]))
    print("wave with the longest tile:", w, "phase sums", [int(v) for v in d[w, :8]], "tiles/tail/mfma-screened", [int(v) for v in d[w, 8:11]], "median wave phases", [int(v) for v in np.median(d[:, :8], axis=0)])
