#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_prof.sh — rocprofv3 kernel stats of the default bench workload (no secondary legs)
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o b -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-extra-legs --no-parity-mode --no-cpu-baseline > "$out/prof_bench.json" 2> "$out/prof_bench.err")
trace=$(find "$out/prof" -name "*kernel_trace.csv" | head -n 1)
python tools/gap_analysis.py "$trace" > "$out/gaps.txt" 2>&1; tail -n 12 "$out/gaps.txt"
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*.db" -delete
stats=$(find "$out/prof" -name "*kernel_stats.csv" | head -n 1)
head -n 45 "$stats" | cut -c1-200
