cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tile in 14 16; do
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_t${tile}_$i -o p -- python3 tools/one_conv.py 8 64 64 640 320 3 $tile > /dev/null 2>&1
done
done
ls gpurun_out | grep pmc_t | head -20
