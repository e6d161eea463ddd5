#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_misc.sh — GroupNorm tests + tile 70 priority experiments
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "groupnorm or partial_sums or group_sums" > "$out/pytest_gn.txt" 2>&1; echo "pytest rc $?"; tail -n 5 "$out/pytest_gn.txt"
timeout 900 python tools/bench_pers_dbg.py > "$out/bench_pers_dbg.txt" 2>&1; cat "$out/bench_pers_dbg.txt"
timeout 600 python tools/bench_gn_fused.py > "$out/bench_gn_fused.txt" 2>&1; cat "$out/bench_gn_fused.txt"
