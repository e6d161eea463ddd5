#!/bin/bash
# round 5, GPU session 3: staged epilogue rows (EPB) — parity, stamps, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > "$out/pytest_ops.txt" 2>&1; echo "pytest ops rc $?"; tail -n 5 "$out/pytest_ops.txt"
MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_stamps.so timeout 300 python tools/stamps.py 19,21,27,26 > "$out/stamps_epb.txt" 2>&1
MFHIP_NO_EPB=1 MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_stamps.so timeout 300 python tools/stamps.py 19,21,27,26 > "$out/stamps_noepb.txt" 2>&1
timeout 600 python tools/bench_tiles.py --tiles 39,48,44,47,49,50,52,43 > "$out/tiles_epb.txt" 2>&1
MFHIP_NO_EPB=1 timeout 600 python tools/bench_tiles.py --tiles 39,48,44,47,49,50,52,43 > "$out/tiles_noepb.txt" 2>&1
for i in 1 2; do
MFHIP_NO_EPB=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile > "$out/bench_noepb$i.json" 2> "$out/bench_noepb$i.err"
timeout 300 python bench.py --steps 4 --warmup 2 --no-parity-mode --no-cpu-baseline --no-profile > "$out/bench_epb$i.json" 2> "$out/bench_epb$i.err"
done
for f in noepb1 epb1 noepb2 epb2; do echo $f; grep -o '"value": [0-9.]*\|"denoise_step": {"ms": [0-9.]*' "$out/bench_$f.json" | head -2; done
grep "epilogue\|===" "$out/stamps_epb.txt" | grep -v "staging wave 0"
echo ---- no epb; grep "epilogue\|===" "$out/stamps_noepb.txt" | grep -v "staging wave 0"
