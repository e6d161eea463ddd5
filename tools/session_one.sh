#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_one.sh <pytest args...> — one pytest invocation on the GPU box
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest "$@" 2>&1 | tail -n 70
