#!/bin/bash
# round 5, GPU session 15: tiles 67 / 68 (two blocks per CU, 2x2 waves of 64x80), idle-gap analysis of the replayed step, re-tune
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -x -q -m gpu > "$out/pytest.txt" 2>&1; echo "pytest rc $?"; tail -n 3 "$out/pytest.txt"
timeout 900 python tools/bench_tiles.py --set conv,lin --tiles 67,68,14,26,20,27,49,54 > "$out/tiles.txt" 2>&1
cut -c1-330 "$out/tiles.txt" | grep -v "^/opt"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o b -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-parity-mode --no-cpu-baseline --no-profile --no-extra-legs > "$out/trace.log" 2>&1)
tr=$(find "$out/trace" -name "*kernel_trace.csv" | head -1); ls -la "$tr"
timeout 300 python tools/gap_analysis.py "$tr" | tee "$out/gaps.txt"
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*.db" -delete
export MFHIP_TUNE_CACHE="$out/user_cache.json"
timeout 2400 python tools/tune_step.py --max-evals 1300 --top 5 --within 0.15 --passes 1 --overlay profiles/r05_tmp/step_tune_s13.json --out "$out/tune_cache_new.json" > "$out/tune_step.txt" 2>&1
grep -v "^/opt\|models built" "$out/tune_step.txt" | cut -c1-220 | tail -n 30
unset MFHIP_TUNE_CACHE
B="--steps 3 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile"
run() { tag=$1; dir=$2; shift 2; (cd $dir && env "$@" timeout 400 python bench.py $B > "$out/b_$tag.json" 2> "$out/b_$tag.err"); echo "$tag: $(grep -o '"value": [0-9.]*' "$out/b_$tag.json" | head -1) $(grep denoise "$out/b_$tag.err" | tail -1 | grep -o 'denoise [0-9.]* ms')"; }
run r04 _r04 MFHIP_TUNE_CACHE=/tmp/none_r04.json
B="$B --no-extra-legs"
cp "$out/tune_cache_new.json" /tmp/tuned.json
run new_tuned . MFHIP_TUNE_CACHE=/tmp/tuned.json
run new . MFHIP_TUNE_CACHE=/tmp/none_new.json
