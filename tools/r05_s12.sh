#!/bin/bash
# round 5, GPU session 12: FX tiles (53-62) + compile-time 16-bit flavour + forced epilogue unrolls: parity, A/B against round 4, re-tune
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -x -q -m gpu > "$out/pytest.txt" 2>&1; echo "pytest rc $?"; tail -n 4 "$out/pytest.txt"
B="--steps 3 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile"
run() { tag=$1; dir=$2; shift 2; (cd $dir && env "$@" timeout 400 python bench.py $B > "$out/b_$tag.json" 2> "$out/b_$tag.err"); echo "$tag: $(grep -o '"value": [0-9.]*' "$out/b_$tag.json" | head -1) $(grep denoise "$out/b_$tag.err" | tail -1 | grep -o 'denoise [0-9.]* ms')"; }
run r04 _r04 MFHIP_TUNE_CACHE=/tmp/none_r04.json
B="$B --no-extra-legs"
run new . MFHIP_TUNE_CACHE=/tmp/none_new.json
run new_r04cache . MFHIP_NO_TUNE_CTX=1 MFHIP_TUNE_CACHE=_r04/reflecting-reality_amd/tune_cache.json
(cd _r04 && timeout 300 python "$GRAFT_REPO_ROOT/tools/ab_ops.py" 2>&1 | grep "groupnorm\|layernorm") > "$out/ab_r04.txt"
timeout 300 python tools/ab_ops.py 2>&1 | grep "groupnorm\|layernorm" > "$out/ab_new.txt"
paste "$out/ab_r04.txt" "$out/ab_new.txt" | cut -c1-150
export MFHIP_TUNE_CACHE="$out/user_cache.json"
timeout 2400 python tools/tune_step.py --max-evals 1300 --top 5 --within 0.15 --passes 1 --overlay reflecting-reality_amd/tune_cache.json --out "$out/tune_cache_new.json" > "$out/tune_step.txt" 2>&1
grep -v "^/opt\|models built" "$out/tune_step.txt" | cut -c1-220 | tail -n 60
unset MFHIP_TUNE_CACHE
run r04b _r04 MFHIP_TUNE_CACHE=/tmp/none_r04.json
cp "$out/tune_cache_new.json" /tmp/tuned.json
run new_tuned . MFHIP_TUNE_CACHE=/tmp/tuned.json
