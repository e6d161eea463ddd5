"""Per-launch averages of the counters collected by tools/pmc_train.sh (algorithmic FLOP; the f16x3 kernels issue 3 MFMAs per product)."""
import collections, csv, glob, sys

root = sys.argv[1]
info = collections.OrderedDict([
    ("wg160", ("weight gradient f16x3, conv3x3 320->320 @64x64 batch 8 (160 x 160 tiles, five waves)", 2.0 * 32768 * 320 * 2880, "conv_wgrad_tr160_kernel")),
    ("wg128", ("weight gradient f16x3, conv3x3 640->640 @32x32 batch 8 (128 x 128 tiles)", 2.0 * 8192 * 640 * 5760, "conv_wgrad_tr_kernel")),
    ("attnbwd_kv", ("flash attention backward, dK / dV pass, d = 40, 4096 x 4096 tokens, 64 (batch, head) pairs (4 products)", 8.0 * 64 * 4096 * 4096 * 40, "attn_bwd_kernel<40, true, false>")),
    ("attnbwd_q", ("flash attention backward, dQ pass (3 products)", 6.0 * 64 * 4096 * 4096 * 40, "attn_bwd_kernel<40, false, false>")),
    ("wg160b", ("weight gradient on bf16 operands (bf16x1 mode), conv3x3 320->320 @64x64 batch 8 (160 x 160 tiles, register-staged)", 2.0 * 32768 * 320 * 2880, "conv_wgrad_tr160_kernel")),
    ("wg128b", ("weight gradient on bf16 operands, conv3x3 640->640 @32x32 batch 8 (128 x 128 tiles, LDS-DMA)", 2.0 * 8192 * 640 * 5760, "conv_wgrad_dma_kernel")),
    ("attnbwdb_kv", ("bf16 flash attention backward, dK / dV pass, d = 40, 4096 x 4096 tokens, 64 (batch, head) pairs (4 products)", 8.0 * 64 * 4096 * 4096 * 40, "attn_bwd_kernel<40, true, true>")),
    ("attnbwdb_q", ("bf16 flash attention backward, dQ pass (3 products)", 6.0 * 64 * 4096 * 4096 * 40, "attn_bwd_kernel<40, false, true>")),
])
DIRS = {"attnbwd_kv": "attnbwd", "attnbwd_q": "attnbwd", "attnbwdb_kv": "attnbwdb", "attnbwdb_q": "attnbwdb"}
for name, (desc, flop, kname) in info.items():
    vals = collections.OrderedDict()
    dur = None
    for d in sorted(glob.glob(f"{root}/pmct_{DIRS.get(name, name)}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if kname in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                vals[k] = sum(v) / len(v)
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            t = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
            if t:
                dur = sum(t) / len(t) / 1e3
    if dur is None:
        print(f"{name}: no data")
        continue
    print(f"{name}: {desc}: {flop / 1e9:.1f} GFLOP algorithmic; kernel duration under the counters {dur:.1f} us = {flop / dur / 1e6:.0f} TF/s")
    for k, v in vals.items():
        print(f"  {k:32s} {v:14.4g}")
    if "GRBM_GUI_ACTIVE" in vals:
        cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
        line = f"  effective clock {cyc / dur / 1e3:.2f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            line += f"; matrix pipe busy {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} of the SIMD cycles"
        if "SQ_BUSY_CYCLES" in vals:
            line += f"; SQ busy {vals['SQ_BUSY_CYCLES'] / (cyc * 8 * 4):.3f} (per XCD-SE units)"
        print(line)
    w = vals.get("SQ_WAVE_CYCLES")
    if w:
        print("  " + ", ".join(f"{k}/WAVE_CYCLES = {vals[k] / w:.3f}" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY",
                                                                                "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC",
                                                                                "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM") if k in vals))
    if "SQ_INSTS_VALU" in vals and vals.get("SQ_INSTS_MFMA"):
        print(f"  VALU instructions per MFMA: {vals['SQ_INSTS_VALU'] / vals['SQ_INSTS_MFMA']:.1f}; LDS instructions per MFMA: "
              f"{vals.get('SQ_INSTS_LDS', 0) / vals['SQ_INSTS_MFMA']:.2f}")
    if "FETCH_SIZE" in vals:
        print(f"  HBM-side traffic per launch: fetch {2 * vals['FETCH_SIZE'] / 1024:.1f} MB (x2 gfx950 correction), write {vals.get('WRITE_SIZE', 0) / 1024:.1f} MB")
    if "TCC_HIT_sum" in vals:
        h, m = vals["TCC_HIT_sum"], vals.get("TCC_MISS_sum", 0)
        print(f"  L2: {h + m:.4g} requests, hit rate {h / max(h + m, 1):.3f}")
    print()
