# rocprofv3 --pmc passes (one counter group per run, kernel trace only) over single launches picked in round 3: the short-K
# 1x1 GEMMs the step spends ~2 ms in, a low-resolution conv, and the d = 40 attention kernel.  Run from the repo root on the GPU
# box:  bash tools/pmc_probe.sh   ->  gpurun_out/pmc_probe.txt
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum")
run() {   # name, program, args...
  name=$1; shift
  i=0
  for c in "${GROUPS_[@]}"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcp_${name}_$i -o p -- python3 "$@" > gpurun_out/pmc_pass.log 2>&1 || echo "[pmc] counter pass $name/$i FAILED (rc=$?): see gpurun_out/pmc_pass.log" >&2
  done
}
run g16_1x1 tools/one_gemm.py bf16 8 16 16 1280 1280 1 44
run g64_1x1 tools/one_gemm.py bf16 8 64 64 320 320 1 26
run g32_1x1 tools/one_gemm.py bf16 8 32 32 640 640 1 48
run c8_3x3 tools/one_gemm.py bf16 8 8 8 1280 1280 3 41
run c32_3x3 tools/one_gemm.py bf16 8 32 32 640 640 3 47
run attn40 tools/one_attn.py
python3 tools/pmc_probe_summary.py gpurun_out > gpurun_out/pmc_probe.txt
cat gpurun_out/pmc_probe.txt
