# rocprofv3 --pmc passes over the dominant conv of the step (3x3 320->320 @ 64x64, B_eff 8) in the bf16 mode (tile 20) and in
# the split-precision parity mode (f16x3, tile 20), and over an fp8 transformer Linear of SDXL (16384 x 640 -> 5120).
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {   # name, args...
  name=$1; shift
  i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcm_${name}_$i -o p -- python3 tools/one_gemm.py "$@" > gpurun_out/pmc_pass.log 2>&1 || echo "[pmc] counter pass FAILED (rc=$?): see gpurun_out/pmc_pass.log" >&2
  done
}
run bf16 bf16 8 64 64 320 320 3 20
run f16x3 f16x3 8 64 64 320 320 3 20
run fp8 fp8 4 64 64 640 5120 1 14
run bf16lin bf16 4 64 64 640 5120 1 14
python3 tools/pmc_modes_summary.py gpurun_out
