#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_pers.sh — the persistent-GEMM tests, where its time goes, and the per-tile table of the short-K GEMMs
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent or transposed_v" > "$out/pytest_pers.txt" 2>&1; echo "pytest rc $?"; tail -n 15 "$out/pytest_pers.txt"
timeout 900 python tools/bench_pers_dbg.py > "$out/bench_pers_dbg.txt" 2>&1; cat "$out/bench_pers_dbg.txt"
timeout 900 python tools/bench_ff1.py > "$out/bench_ff1.txt" 2>&1; cat "$out/bench_ff1.txt"
