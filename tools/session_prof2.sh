#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_prof2.sh — the range-guard tests, then tools/session_prof.sh
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_training_gpu.py -x -q -m gpu -k "guard or overflow or flag_raised" 2>&1 | tail -n 8
bash tools/session_prof.sh
