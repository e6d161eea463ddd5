#!/bin/bash
# round 5, GPU session 4: isolated re-tune with the new tiles + whole-step tuning
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
export MFHIP_TUNE_CACHE="$out/user_cache.json"
timeout 2400 python tools/tune_step.py --max-evals 160 --out "$out/tune_cache_new.json" > "$out/tune_step.txt" 2>&1
cp gpurun_out/tune_rankings.json "$out/" 2>/dev/null
tail -n 40 "$out/tune_step.txt"
