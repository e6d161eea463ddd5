#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_gn.sh — the round-6 GroupNorm tests + A/B microbenchmark
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -x -q -m gpu -k "groupnorm or partial_sums or refused" > "$out/pytest_gn.txt" 2>&1; echo "pytest rc $?"; tail -n 15 "$out/pytest_gn.txt"
timeout 600 python tools/bench_gn_fused.py > "$out/bench_gn_fused.txt" 2>&1; cat "$out/bench_gn_fused.txt"
