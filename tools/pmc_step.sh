#!/bin/bash
# HBM-side traffic of the mf_gemm_conv family over one denoise step: two separate rocprofv3 --pmc passes
# (plus two passes of the L2 -> fabric request counters; FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"); summarise with
# tools/pmc_step_summary.py.  Run from the repo root on the GPU box:  bash tools/pmc_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE EA_RD EA_WR; do
  ctr=$c
  # the L2 -> fabric request counters: all requests / the 32-byte (64-byte) ones / those destined for the memory controllers
  [ $c = EA_RD ] && ctr="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum"
  [ $c = EA_WR ] && ctr="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_step_$c -o step -- python3 tools/profile_step.py --steps 1 > gpurun_out/pmc_step_$c.log 2>&1
done
python3 tools/pmc_step_summary.py gpurun_out/pmc_step_FETCH_SIZE gpurun_out/pmc_step_WRITE_SIZE gpurun_out/pmc_step_EA_RD gpurun_out/pmc_step_EA_WR > gpurun_out/pmc_gemm_family.json
cat gpurun_out/pmc_gemm_family.json
