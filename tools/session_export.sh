#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_export.sh — full-size step program: export, C host, comparison
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
df -h /tmp | tail -n 1
timeout 2400 python tools/export_step.py "$@" 2>&1 | tail -n 30
