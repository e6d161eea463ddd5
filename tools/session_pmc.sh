#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_pmc.sh — counter passes of the GEMM family (tools/pmc_step.sh) and of the d = 40 attention (tools/pmc_attn.sh)
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
bash tools/pmc_step.sh > "$out/pmc_step.txt" 2>&1; cp gpurun_out/pmc_gemm_family.json "$out/" 2>/dev/null; cat "$out/pmc_gemm_family.json"
bash tools/pmc_attn.sh > "$out/pmc_attn_d40.txt" 2>&1; tail -n 30 "$out/pmc_attn_d40.txt"
