#!/bin/bash
# One gpurun lease: `gpurun --timeout N -- 'bash tools/gpu_session.sh <tag> <cmd...>'` runs <cmd...> from the repo root with its
# output under gpurun_out/<tag>/ (merged back by gpurun).  Hardened as ADVICE r3 asked: no unset-variable cd, no relative rm.
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
tag="${1:?usage: gpu_session.sh <tag> <command...>}"; shift
out="$GRAFT_REPO_ROOT/gpurun_out/$tag"
mkdir -p "$out"
export MF_SESSION_OUT="$out" TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
"$@" > "$out/log.txt" 2>&1
rc=$?
echo "exit code $rc" >> "$out/log.txt"
tail -n 60 "$out/log.txt"
exit $rc
