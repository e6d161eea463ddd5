"""Summarise a rocprofv3 kernel trace CSV: per-kernel totals for the LAST denoise step, sum vs wall span."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "timestep_embedding" in r["Kernel_Name"]]
s = idx[-2] if len(idx) >= 2 else 0
agg = collections.defaultdict(lambda: [0, 0.0])
t0 = int(rows[s]["Start_Timestamp"])
t1 = int(rows[-1]["End_Timestamp"])
gaps = 0.0
prev = None
for r in rows[s:]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    nm = re.sub(r"\(.*", "", nm)
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[nm][0] += 1
    agg[nm][1] += (b - a) / 1e3
    if prev is not None and a > prev:
        gaps += (a - prev) / 1e3
    prev = max(prev or 0, b)
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"{v[1] / 1e3:8.3f} ms {100 * v[1] / tot:5.1f}% n={v[0]:4d} avg={v[1] / v[0]:7.1f}us {k[:84]}")
print(f"sum kernels {tot / 1e3:.3f} ms | wall span {(t1 - t0) / 1e6:.3f} ms | idle gaps {gaps / 1e3:.3f} ms | launches {sum(v[0] for v in agg.values())}")
