# rocprofv3 --pmc passes (one counter group per run, kernel-trace only) over the flash attention kernel at the dominant
# shape of the step: self-attention, 8 heads x d = 40, 4096 tokens, B_eff = 8.  Run from the repo root on the GPU box.
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmca_$i -o p -- python3 tools/one_attn.py > gpurun_out/pmc_pass.log 2>&1 || echo "[pmc] counter pass FAILED (rc=$?): see gpurun_out/pmc_pass.log" >&2
done
python3 tools/pmc_attn_summary.py gpurun_out
