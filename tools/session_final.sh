#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_final.sh — the round's closing session: every GPU test, smoke(), the default bench line, kernel stats
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/session_tests_bench.sh
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
bash tools/session_prof.sh
