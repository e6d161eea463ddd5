"""Join gpurun_out/shape_seq.json (tools/gemm_shapes.py) with a rocprofv3 kernel trace of the same run: per-shape
GPU-side durations of the mf_gemm_conv launches (event timing in eager mode is inflated for short kernels).

    rocprofv3 --kernel-trace -d gpurun_out/profN -o step -- python3 tools/gemm_shapes.py 4
    python tools/join_trace.py gpurun_out/profN/step_kernel_trace.csv gpurun_out/shape_seq.json
"""
import collections
import csv
import json
import sys

if sys.argv[1].endswith(".db"):         # rocprofv3's default rocpd output
    import sqlite3
    rows = [{"Kernel_Name": n, "Start_Timestamp": a, "End_Timestamp": b} for n, a, b in
            sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels") if "gemm_conv_kernel" in n]
else:
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_conv_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = json.load(open(sys.argv[2]))
rows = rows[-len(seq):]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
tot = 0.0
for r, (f, k) in zip(rows, seq):
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    a = agg[tuple(k)]
    a[0] += 1; a[1] += t; a[2] += f
    tot += t
print(f"{len(seq)} launches, {tot * 1e3:.2f} ms GPU time, {sum(f for f, _ in seq) / tot / 1e12:.1f} TFLOP/s overall")
print(f"{'M':>7} {'N':>6} {'K':>6} kh s u nz tile sk |  n   total_us  avg_us  TF/s  %time")
for k, (c, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 50]:
    print(f"{k[0]:7d} {k[1]:6d} {k[2]:6d} {k[3]:2d} {k[4]} {k[5]} {k[6]:2d} {k[7]:4d} {k[8]:2d} | {c:3d} {t * 1e6:9.1f} {t / c * 1e6:7.1f} {f / t / 1e12:6.1f} {100 * t / tot:5.1f}")
