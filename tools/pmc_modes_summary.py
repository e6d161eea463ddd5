"""Per-launch averages of the counters collected by tools/pmc_modes.sh (gemm_conv_kernel launches only)."""
import collections, csv, glob, sys

root = sys.argv[1]
info = {"bf16": ("conv3x3 320->320 @64x64, B_eff 8, bf16, tile 20", 2.0 * 32768 * 320 * 2880, 1),
        "f16x3": ("the same conv in split precision f16x3 (three fp16 MFMAs per product), tile 20", 2.0 * 32768 * 320 * 2880, 3),
        "fp8": ("fp8 Linear 16384 x 640 -> 5120 (SDXL FF), e4m3 operands, tile 14", 2.0 * 16384 * 640 * 5120, 1),
        "bf16lin": ("the same Linear in bf16, tile 14", 2.0 * 16384 * 640 * 5120, 1)}
for name, (desc, flop, nm) in info.items():
    vals = collections.OrderedDict()
    dur = None
    for d in sorted(glob.glob(f"{root}/pmcm_{name}_*")):
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
    if dur is None:
        print(f"{name}: no data")
        continue
    print(f"{name}: {desc}: {flop / 1e9:.1f} GFLOP algorithmic; kernel duration under the counters {dur:.1f} us = {flop / dur / 1e6:.0f} TF/s")
    for k, v in vals.items():
        print(f"  {k:32s} {v:14.4g}")
    if "GRBM_GUI_ACTIVE" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
        print(f"  effective clock {cyc / dur / 1e3:.2f} GHz; matrix pipe busy {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} of the SIMD cycles")
    w = vals.get("SQ_WAVE_CYCLES")
    if w:
        print("  " + ", ".join(f"{k}/WAVE_CYCLES = {vals[k] / w:.3f}" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in vals))
    if "FETCH_SIZE" in vals:
        print(f"  HBM-side traffic per launch: fetch {2 * vals['FETCH_SIZE'] / 1024:.1f} MB (x2 gfx950 correction), write {vals.get('WRITE_SIZE', 0) / 1024:.1f} MB")
