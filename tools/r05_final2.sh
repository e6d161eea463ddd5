#!/bin/bash
# round 5, closing session: re-tune with tile 69 on offer, every GPU test, smoke(), the default bench line, kernel stats of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
export MFHIP_TUNE_CACHE="$out/user_cache.json"
timeout 2400 python tools/tune_step.py --max-evals 900 --top 5 --within 0.15 --passes 1 --overlay reflecting-reality_amd/tune_cache.json --out "$out/tune_cache_new.json" > "$out/tune_step.txt" 2>&1
grep -v "^/opt\|models built" "$out/tune_step.txt" | cut -c1-220 | tail -n 20
unset MFHIP_TUNE_CACHE
B="--steps 3 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile"
run() { tag=$1; dir=$2; shift 2; (cd $dir && env "$@" timeout 400 python bench.py $B > "$out/b_$tag.json" 2> "$out/b_$tag.err"); echo "$tag: $(grep -o '"value": [0-9.]*' "$out/b_$tag.json" | head -1) $(grep denoise "$out/b_$tag.err" | tail -1 | grep -o 'denoise [0-9.]* ms')"; }
run r04 _r04 MFHIP_TUNE_CACHE=/tmp/none_r04.json
B="$B --no-extra-legs"
cp "$out/tune_cache_new.json" /tmp/tuned.json
run new_tuned . MFHIP_TUNE_CACHE=/tmp/tuned.json
run new . MFHIP_TUNE_CACHE=/tmp/none_new.json
timeout 3000 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "pytest -m gpu rc $?"; tail -n 2 "$out/pytest_gpu.txt"
