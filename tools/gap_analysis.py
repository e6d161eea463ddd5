"""Idle time between kernels of the replayed denoise step, from a rocprofv3 kernel trace (…_kernel_trace.csv):
how much of the denoise loop has NO kernel running, exactly one, or two and more (the two streams of the step graph)."""
import csv
import sys
from collections import Counter

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
rows.sort()
print(f"{len(rows)} kernel records, span {(rows[-1][1] - rows[0][0]) / 1e6:.1f} ms")
# the denoise loops: the longest stretches of back-to-back GEMM-family kernels; take the window between the first and last cfg_ddim kernel of
# the LAST pass (50 of them per pass)
ddim = [i for i, r in enumerate(rows) if "cfg_ddim" in r[2] or "ddim" in r[2].lower()]
if len(ddim) >= 50:
    lo, hi = ddim[-50], ddim[-1]
else:
    lo, hi = 0, len(rows) - 1
win = rows[lo:hi + 1]
t0, t1 = win[0][0], win[-1][1]
ev = []
for s, e, *_ in win:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
lvl, last = 0, t0
time_at = Counter()
for t, d in ev:
    time_at[min(lvl, 3)] += t - last
    last = t
    lvl += d
tot = t1 - t0
steps = 49.0
print(f"window: {len(win)} kernels over {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.3f} ms per step, {len(win) / steps:.0f} kernels per step")
for k in sorted(time_at):
    print(f"  {k}{'+' if k == 3 else ' '} kernels running: {time_at[k] / 1e6:8.2f} ms  {100.0 * time_at[k] / tot:5.1f} %   ({time_at[k] / 1e3 / steps:7.1f} us per step)")
# gaps where nothing runs: distribution
gaps = []
lvl, last_end = 0, None
for t, d in ev:
    if d == 1 and lvl == 0 and last_end is not None:
        gaps.append(t - last_end)
    lvl += d
    if lvl == 0:
        last_end = t
gaps.sort()
if gaps:
    n = len(gaps)
    print(f"  idle gaps: {n} ({n / steps:.0f} per step), median {gaps[n // 2] / 1e3:.2f} us, 90 % {gaps[int(n * 0.9)] / 1e3:.2f} us, max {gaps[-1] / 1e3:.1f} us, sum {sum(gaps) / 1e3 / steps:.1f} us per step")
# per-kernel-family busy time in the window
fam = Counter()
for s, e, name, *_ in win:
    key = name.split("<")[0].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").replace("mfgemm::", "")
    fam[key] += e - s
for k, v in fam.most_common(12):
    print(f"  {k:40s} {v / 1e3 / steps:9.1f} us per step")
