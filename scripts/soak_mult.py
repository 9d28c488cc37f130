"""Dev helper: 3000 steady-state steps of the Multinomial path at C4 size (N=1e6, D=1000, K=32) with the device master (Dirichlet draws
launched ahead, rows on demand): time per step must stay flat, device memory must not grow, the labels must stay where they are, and
nearly every step must take the draws made ahead."""
import sys, time, importlib, json
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
host = importlib.import_module("dpmmsubclusters_jl_amd.host")
N, D, K = 10 ** 6, 1000, 32
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(0)
P = rng.dirichlet(np.ones(D) * 0.5, size=K)
y = rng.integers(1, K + 1, N)
X = np.empty((N, D), np.float32)
for k in range(K):
    m = y == k + 1
    X[m] = rng.multinomial(100, P[k], size=int(m.sum()))
prior = host.multinomial_hyper(np.ones(D, np.float32))
wk = pkg.Worker(pkg.PRIOR_MULT, D, N, device=0, seed=1)
wk.upload_points(X)
s = host.DPMMSampler(wk, prior, 10.0, N, 1, burnout=20)
s.start_from_labels(y, 1 + np.random.default_rng(0).integers(0, 2, N), K)
def vram():
    try:
        import torch
        free, total = torch.cuda.mem_get_info(0)
        return (total - free) / 2 ** 20
    except Exception:
        return -1.0
ts, m0, marks, kch = [], None, [], []
for i in range(steps):
    k0 = s.K
    t0 = time.perf_counter(); s.group_step(False, False); ts.append(time.perf_counter() - t0)
    if s.K != k0: kch.append((i, k0, s.K))
    if i == 100: m0 = vram()
    if i in (200, 500) or (i > 0 and i % 1000 == 0): marks.append((i, vram()))
m1 = vram()
ts = np.array(ts) * 1e3
lab, _ = wk.get_labels()
from dpmmsubclusters_jl_amd.host.sampler import nmi_vi_from_contingency
C = np.zeros((int(lab.max()), int(y.max())))
np.add.at(C, (lab - 1, y - 1), 1)
nmi = float(nmi_vi_from_contingency(C)[0])
print(json.dumps(dict(steps=steps, ms_first500=float(ts[100:600].mean()), ms_last500=float(ts[-500:].mean()), ms_max_after_100=float(ts[100:].max()),
                      K_final=int(s.K), vram_mib_after_100=m0, vram_mib_at_end=m1, vram_mib_marks=marks, label_agreement=float((lab == y).mean()), nmi_vs_generator=nmi, K_changes=kch,
                      draws_taken_from_the_set_made_ahead=wk.debug_mult_draws_ahead())))
wk.close()
