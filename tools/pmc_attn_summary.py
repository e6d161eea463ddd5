"""Per-launch averages of the counters collected by tools/pmc_attn.sh (attn_fwd_kernel launches only)."""
import collections, csv, glob, sys

root = sys.argv[1]
vals = collections.OrderedDict()
dur = None
for d in sorted(glob.glob(f"{root}/pmca_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "attn_fwd" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        t = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "attn_fwd" in r["Kernel_Name"]]
        if t:
            dur = sum(t) / len(t) / 1e3
flop = 4.0 * 8 * 8 * 4096 * 4096 * 40
print(f"attention kernel (d = 40): self-attention B_eff 8 x 8 heads x 4096 tokens x d 40 ({flop / 1e9:.1f} GFLOP); kernel duration under the counters "
      f"{dur:.1f} us = {flop / dur / 1e6:.0f} TF/s")
for k, v in vals.items():
    print(f"  {k:32s} {v:14.4g}")
w = vals.get("SQ_WAVE_CYCLES")
if w:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if k in vals:
            print(f"  {k} / SQ_WAVE_CYCLES = {vals[k] / w:.3f}")
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    # SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over SIMDs (32 per 32x32x16 MFMA); GRBM_GUI_ACTIVE is summed over the
    # 8 XCDs (MI355X_MICROARCH.md, DVFS give-back): kernel cycles = GRBM / 8, available matrix-pipe cycles = that x 256 CUs x 4 SIMDs
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
    print(f"  effective clock {cyc / dur / 1e3:.2f} GHz; matrix pipe busy {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} of the kernel's "
          f"SIMD cycles ({vals['SQ_INSTS_MFMA']:.3g} MFMAs x 32 cycles)")
if "SQ_INSTS_VALU" in vals and "SQ_INSTS_MFMA" in vals:
    print(f"  VALU instructions per MFMA: {vals['SQ_INSTS_VALU'] / vals['SQ_INSTS_MFMA']:.1f}")
if "FETCH_SIZE" in vals:
    print(f"  HBM-side traffic per launch: fetch {2 * vals['FETCH_SIZE'] / 1024:.1f} MB (FETCH_SIZE KB x 2, gfx950 wide-read correction), "
          f"write {vals.get('WRITE_SIZE', 0) / 1024:.1f} MB; algorithmic q + k + v^T + out = {4 * 8 * 4096 * 320 * 2 / 1e6:.1f} MB")
