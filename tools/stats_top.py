"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: python tools/stats_top.py <dir> [n] [divide_by]"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / div:.2f} ms (/{div:g})")
for r in rows[:n]:
    print(f"{r['Name'][:118]:118s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6 / div:9.2f} ms {100 * float(r['TotalDurationNs']) / tot:5.1f}% {float(r['AverageNs']) / 1e3:9.1f} us")
