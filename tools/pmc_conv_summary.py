"""Per-launch averages of the counters collected by tools/pmc_conv.sh (gemm_conv_kernel launches only)."""
import collections, csv, glob, sys

root = sys.argv[1]
for tile in (20, 14):
    vals = collections.OrderedDict()
    dur = None
    for d in sorted(glob.glob(f"{root}/pmcc_t{tile}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "gemm_conv_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                vals[k] = sum(v) / len(v)
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            t = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "gemm_conv_kernel" in r["Kernel_Name"]]
            if t:
                dur = sum(t) / len(t) / 1e3
    print(f"tile {tile}: conv3x3 320->320 @64x64, B_eff 8 (M=32768 N=320 K=2880, 60.4 GFLOP); kernel duration under the counters {dur:.1f} us")
    for k, v in vals.items():
        print(f"  {k:32s} {v:14.4g}")
