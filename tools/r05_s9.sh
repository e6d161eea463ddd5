#!/bin/bash
# round 5, GPU session 9: the division-free prologue — parity of every GEMM test, stamps, bench
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_split_gpu.py tests/test_layers_gpu.py -x -q -m gpu > "$out/pytest.txt" 2>&1; echo "pytest rc $?"; tail -n 5 "$out/pytest.txt"
MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_stamps.so timeout 300 python tools/stamps.py 19,21,26,27,4,5 > "$out/stamps.txt" 2>&1
grep "===\|prologue\|tile 0 landed\|launch span" "$out/stamps.txt" | grep -v "^   -> main"
for i in 1 2; do
timeout 400 python bench.py --steps 4 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile --no-extra-legs > "$out/bench$i.json" 2> "$out/bench$i.err"
grep -o '"value": [0-9.]*' "$out/bench$i.json" | head -1; grep "denoise" "$out/bench$i.err" | tail -1
(cd _r04 && MFHIP_TUNE_CACHE=/tmp/none.json timeout 400 python bench.py --steps 4 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile > "$out/bench_r04_$i.json" 2> "$out/bench_r04_$i.err")
grep -o '"value": [0-9.]*' "$out/bench_r04_$i.json" | head -1; grep "denoise" "$out/bench_r04_$i.err" | tail -1
done
