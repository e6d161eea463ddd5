#!/bin/bash
# round 5, GPU session 2: who waits for whom in the warp-specialised main loops (stamped build)
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_stamps.so timeout 600 python tools/stamps.py 7,8,9,19,20,21,22,23,24,25,26,27 > "$out/stamps.txt" 2>&1
cat "$out/stamps.txt"
