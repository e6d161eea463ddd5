#!/bin/bash
# round 5, GPU session 8: whole-step tuning of the parity mode (f16x3) and the fp16 mode; f16x3 stamps; counter list
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
export MFHIP_TUNE_CACHE="$out/user_cache.json"
MF_STAMPS_PREC=f16x3 MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_stamps.so timeout 600 python tools/stamps.py 28,29,30,31,32 > "$out/stamps_f16x3.txt" 2>&1
grep -v "^   ->\|block entry\|epilogue round\|staging wave 0: barrier" "$out/stamps_f16x3.txt"
timeout 1500 python tools/tune_step.py --precision f16x3 --max-evals 300 --top 4 --within 0.10 --out "$out/tune_f16x3.json" > "$out/tune_f16x3.txt" 2>&1
grep -v "^/opt\|models built" "$out/tune_f16x3.txt" | cut -c1-220 | tail -n 25
timeout 1500 python tools/tune_step.py --precision fp16 --max-evals 400 --top 4 --within 0.12 --out "$out/tune_fp16.json" > "$out/tune_fp16.txt" 2>&1
grep -v "^/opt\|models built" "$out/tune_fp16.txt" | cut -c1-220 | tail -n 25
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > "$out/counters.txt" 2>&1; grep -i -c "" "$out/counters.txt"; grep -i "dram\|mall\|hbm\|EA0_RDREQ\|EA0_WRREQ" "$out/counters.txt" | head -40
