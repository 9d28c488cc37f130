"""Kernel timeline of ONE steady-state step from a rocprofv3 kernel trace (csv): start offset, duration and the gap in front of every
kernel between two consecutive sweep launches (the step of median length among the last 21 complete steps of the trace), plus totals.
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 scripts/config_step.py niw 64 1250000 100
   python3 scripts/step_timeline.py gpurun_out/tl [sweep-kernel-substring]"""
import csv, glob, os, sys
root = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "sweep"
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dpmm::", ""), r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
sw = [i for i, r in enumerate(rows) if key in r[2] and "pack" not in r[2]]
# the step of MEDIAN length among the last 21 complete ones (one step alone can be a host hiccup: the gaps are the host's)
cand = [(rows[sw[i + 1]][0] - rows[sw[i]][0], i) for i in range(max(0, len(sw) - 23), len(sw) - 2)]
cand.sort()
mid = cand[len(cand) // 2][1] if cand else len(sw) - 3
a, b = sw[mid], sw[mid + 1]
t0 = rows[a][0]
prev_end = None
busy = 0
print(f"step of {(rows[b][0] - t0) / 1e3:.1f} us, {b - a} kernels")
for s, e, n, q in rows[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  q{q}  {n[:90]}")
    prev_end = max(prev_end, e) if prev_end is not None else e
    busy += e - s
print(f"sum of kernel durations {busy / 1e3:.1f} us")
# averages over all steps
from collections import defaultdict
tot = defaultdict(lambda: [0, 0])
for s, e, n, q in rows[sw[len(sw) // 2]:sw[-1]]:
    tot[n][0] += e - s; tot[n][1] += 1
nsteps = len(sw) - 1 - len(sw) // 2
print(f"per-step averages over the last {nsteps} steps:")
for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {t / 1e3 / nsteps:8.1f} us/step  {c / nsteps:5.2f} launches/step  {n[:90]}")
