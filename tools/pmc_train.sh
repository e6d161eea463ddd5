# rocprofv3 --pmc passes (one counter group per run, kernel trace only) over the training step's two heaviest non-GEMM kernels:
# the f16x3 weight gradient (both tile forms) and the flash attention backward.  Run from the repo root on the GPU box:
#   bash tools/pmc_train.sh   ->  gpurun_out/pmc_train.txt
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum")
run() {   # name, program, args...
  name=$1; shift
  i=0
  for c in "${GROUPS_[@]}"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmct_${name}_$i -o p -- python3 "$@" > gpurun_out/pmc_pass.log 2>&1 || echo "[pmc] counter pass $name/$i FAILED (rc=$?): see gpurun_out/pmc_pass.log" >&2
  done
}
run wg160 tools/one_wgrad.py 8 64 320 320 3
run wg128 tools/one_wgrad.py 8 32 640 640 3
run attnbwd tools/one_attn_bwd.py
run wg160b tools/one_wgrad.py 8 64 320 320 3 bf16
run wg128b tools/one_wgrad.py 8 32 640 640 3 bf16
run attnbwdb tools/one_attn_bwd.py bf16x1
python3 tools/pmc_train_summary.py gpurun_out > gpurun_out/pmc_train.txt
cat gpurun_out/pmc_train.txt
