#!/bin/bash
# gpurun: bash tools/gpu_session.sh <tag> bash tools/session_prio.sh — static wave priority in the warp-specialised conv tiles (MFHIP_DBG_EPI bits 16 / 32)
: "${GRAFT_REPO_ROOT:?run through gpurun}"; : "${MF_SESSION_OUT:?run through tools/gpu_session.sh}"
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
for bits in 0 16 32 0; do
  echo "== MFHIP_DBG_EPI=$bits" | tee -a "$out/prio.txt"
  MFHIP_DBG_EPI=$bits timeout 600 python tools/bench_tiles.py --set conv,up --tiles 68,49,54,62,50 2>&1 | grep -v amdgpu.ids | tee -a "$out/prio.txt" | cut -c1-200
done
