#!/bin/bash
# tile 69 (persistent short-K GEMM): parity, then per-tile timings of the transformer block's Linears
cd "$GRAFT_REPO_ROOT" || exit 1
out="$MF_SESSION_OUT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent_short_k" > "$out/pytest.txt" 2>&1; echo "pytest rc $?"; tail -n 25 "$out/pytest.txt" | cut -c1-250
timeout 600 python tools/bench_ff1.py bf16 2>&1 | grep -v "^/opt" | tee "$out/ff1.txt"
