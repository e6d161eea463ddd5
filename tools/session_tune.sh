#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_tune.sh — whole-step tile tuning (tools/tune_step.py) with the current kernels
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 2400 python tools/tune_step.py --max-evals 400 --passes 2 --out "$out/tune_cache_new.json" > "$out/tune_step.txt" 2>&1; echo "tune_step rc $?"
tail -n 40 "$out/tune_step.txt" | cut -c1-200
