#!/bin/bash
# counter passes over ONE launch shape (tools/one_nloop.py <tile>): where do the cycles of the persistent short-K GEMM go?
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tile in 69 29; do
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/p${tile}_$i" -o p -- python3 tools/one_nloop.py $tile > "$out/pass.log" 2>&1 || echo "pass failed: $c"
done
python3 - "$out" $tile <<'PY'
import csv, glob, sys
out, tile = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(f"{out}/p{tile}_*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gemm_nloop" in r["Kernel_Name"] or "gemm_conv_kernel" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(f"== tile {tile}")
for k in sorted(vals):
    print(f"  {k:28s} {vals[k]:.4g}")
w = vals.get("SQ_WAVE_CYCLES", 1.0)
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
    if k in vals:
        print(f"  {k} / SQ_WAVE_CYCLES = {vals[k] / w:.3f}")
PY
done
