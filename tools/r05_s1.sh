#!/bin/bash
# round 5, GPU session 1: the split build + cross-tile fragment pipeline against the round-4 library, on one box
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
R04=reflecting-reality_amd/lib/libmfhip_r04.so
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > "$out/pytest_ops.txt" 2>&1; echo "pytest ops rc $?"; tail -n 3 "$out/pytest_ops.txt"
MFHIP_LIB=$R04 timeout 600 python tools/bench_tiles.py --tiles 39,42,46,48,44,41,47 > "$out/tiles_r04.txt" 2>&1
timeout 900 python tools/bench_tiles.py --tiles 39,42,46,48,44,41,47,49,50,51,52,40,43,45 > "$out/tiles_new.txt" 2>&1
MFHIP_LIB=$R04 timeout 300 python bench.py --steps 4 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile > "$out/bench_r04.json" 2> "$out/bench_r04.err"
timeout 300 python bench.py --steps 4 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile > "$out/bench_new.json" 2> "$out/bench_new.err"
grep -o '"value": [0-9.]*\|"denoise_step": {"ms": [0-9.]*' "$out/bench_r04.json" | head -3
grep -o '"value": [0-9.]*\|"denoise_step": {"ms": [0-9.]*' "$out/bench_new.json" | head -3
cat "$out/tiles_new.txt"
