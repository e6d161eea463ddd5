#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_tests_bench.sh — every GPU test (timed), then the default bench line
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
t0=$(date +%s)
timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 > "$out/pytest_gpu.txt" 2>&1; echo "pytest -m gpu rc $? in $(( $(date +%s) - t0 )) s"; tail -n 40 "$out/pytest_gpu.txt"
timeout 1800 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc $?"; tail -n 3 "$out/bench_default.err"; cat "$out/bench_default.json"
