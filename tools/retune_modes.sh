#!/bin/bash
# whole-step re-tune of the fp16 and f16x3 (parity) modes on the current library (tools/tune_step.py); winners land in gpurun_out/<session>/
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
for prec in fp16 f16x3; do
  export MFHIP_TUNE_CACHE="$out/user_cache_$prec.json"
  timeout 1500 python tools/tune_step.py --precision $prec --max-evals 700 --top 4 --within 0.12 --passes 1 --out "$out/tune_cache_$prec.json" > "$out/tune_step_$prec.txt" 2>&1
  grep -v "^/opt\|models built" "$out/tune_step_$prec.txt" | cut -c1-200 | tail -n 14
  unset MFHIP_TUNE_CACHE
done
