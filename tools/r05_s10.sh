#!/bin/bash
# round 5, GPU session 10: why is the whole pass slower than round 4 on the same box?  bisect + kernel stats; early A-window parity
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
B="--steps 3 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 400 python bench.py $B > "$out/b_$tag.json" 2> "$out/b_$tag.err"; echo "$tag: $(grep -o '"value": [0-9.]*' "$out/b_$tag.json" | head -1) $(grep denoise "$out/b_$tag.err" | tail -1 | grep -o 'denoise [0-9.]* ms')"; }
run new X=1
(cd _r04 && MFHIP_TUNE_CACHE=/tmp/none.json timeout 400 python bench.py --steps 3 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile > "$out/b_r04.json" 2> "$out/b_r04.err"); echo "r04: $(grep -o '"value": [0-9.]*' "$out/b_r04.json" | head -1) $(grep denoise "$out/b_r04.err" | tail -1 | grep -o 'denoise [0-9.]* ms')"
run new_r04cache MFHIP_NO_TUNE_CTX=1 MFHIP_TUNE_CACHE=_r04/reflecting-reality_amd/tune_cache.json
run new_noctx MFHIP_NO_TUNE_CTX=1
run new_noepb MFHIP_NO_EPB=1
run noxt MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_noxt.so
run noxt_r04cache MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_noxt.so MFHIP_NO_TUNE_CTX=1 MFHIP_TUNE_CACHE=_r04/reflecting-reality_amd/tune_cache.json MFHIP_NO_EPB=1
run new2 X=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -x -q -m gpu > "$out/pytest.txt" 2>&1; echo "pytest rc $?"; tail -n 4 "$out/pytest.txt"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_new" -o b -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-parity-mode --no-cpu-baseline --no-profile --no-extra-legs > "$out/prof_new.log" 2>&1
cd "$GRAFT_REPO_ROOT/_r04" && MFHIP_TUNE_CACHE=/tmp/none.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_r04" -o b -- python3 bench.py --steps 2 --warmup 1 --no-parity-mode --no-cpu-baseline --no-profile > "$out/prof_r04.log" 2>&1
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*.db" -delete
ls "$out"/prof_new/* "$out"/prof_r04/* | head
