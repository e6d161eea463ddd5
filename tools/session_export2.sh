#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_export2.sh — program tests, then the full-size step program from a C host
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_program_gpu.py -x -q -m gpu -s 2>&1 | tail -n 40
timeout 2400 python tools/export_step.py "$@" 2>&1 | tail -n 14
