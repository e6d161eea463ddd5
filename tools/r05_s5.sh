#!/bin/bash
# round 5, GPU session 5: longer whole-step search (top 5 within 15 %, two sweeps), continuing from session 4's winners
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
export MFHIP_TUNE_CACHE="$out/user_cache.json"
timeout 2400 python tools/tune_step.py --max-evals 900 --top 5 --within 0.15 --passes 2 --overlay profiles/r05_tmp/step_tune_s4.json --out "$out/tune_cache_new.json" > "$out/tune_step.txt" 2>&1
cp gpurun_out/tune_rankings.json "$out/" 2>/dev/null
tail -n 40 "$out/tune_step.txt"
