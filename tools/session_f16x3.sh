#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_f16x3.sh — the bf16x1 guard tests, then rocprofv3 kernel stats of the parity mode's step
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 1200 python -m pytest tests/test_training_gpu.py -x -q -m gpu -k "guard or overflow or flag_raised" 2>&1 | tail -n 8
(cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o b -- python3 "$GRAFT_REPO_ROOT/bench.py" --precision f16x3 --no-extra-legs --no-parity-mode --no-cpu-baseline > "$out/prof_bench.json" 2> "$out/prof_bench.err")
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*.db" -delete
stats=$(find "$out/prof" -name "*kernel_stats.csv" | head -n 1)
cp "$stats" "$out/f16x3_kernel_stats.csv"
head -n 50 "$stats" | cut -c1-230
tail -n 3 "$out/prof_bench.json" | cut -c1-1500
