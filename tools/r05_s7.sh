#!/bin/bash
# round 5, GPU session 7: parity-mode (f16x3) per-tile data + position-tagged whole-step tuning
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 900 python tools/bench_tiles.py --prec f16x3 --set conv,lin --tiles 14,20,21,37,38,41,44,1,7 > "$out/tiles_f16x3.txt" 2>&1
MF_STAMPS_PREC=f16x3 MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_stamps.so timeout 600 python tools/stamps.py 28,29,30,31,32 > "$out/stamps_f16x3.txt" 2>&1
grep -v "^   ->\|block entry\|epilogue round\|staging wave 0: barrier" "$out/stamps_f16x3.txt"
export MFHIP_TUNE_CACHE="$out/user_cache.json"
timeout 2400 python tools/tune_step.py --max-evals 900 --top 4 --within 0.12 --passes 1 --overlay profiles/r05_tmp/step_tune_s5.json --out "$out/tune_cache_new.json" > "$out/tune_step.txt" 2>&1
grep -v "^/opt\|models built" "$out/tune_step.txt" | cut -c1-220 | tail -n 40
cat "$out/tiles_f16x3.txt" | cut -c1-400
