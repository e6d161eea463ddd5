# rocprofv3 --pmc passes (one counter group per run, kernel-trace only) over the dominant conv of the denoise step:
# conv3x3 320->320 @ 64x64, B_eff = 8, with the dx-tap-reuse tile the tuner picks (20) and the plain tile (14).
# Summarise with tools/pmc_conv_summary.py.  Run from the repo root on the GPU box:  bash tools/pmc_conv.sh
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tile in 20 14; do
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcc_t${tile}_$i -o p -- python3 tools/one_conv.py 8 64 64 320 320 3 $tile > gpurun_out/pmc_pass.log 2>&1 || echo "[pmc] counter pass FAILED (rc=$?): see gpurun_out/pmc_pass.log" >&2
done
done
python3 tools/pmc_conv_summary.py gpurun_out
