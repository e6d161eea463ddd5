#!/bin/bash
# SDXL + BrushNet-XL (BASELINE.json configs[4]) for the round's record: bf16 and fp8 Linears.
O=$MF_SESSION_OUT
timeout 1500 python bench.py --model sdxl --no-cpu-baseline --no-parity-mode 2>&1 | grep -a '"metric"' > $O/bench_sdxl_bf16.json; cut -c1-260 $O/bench_sdxl_bf16.json
timeout 1500 python bench.py --model sdxl --precision fp8 --no-cpu-baseline --no-parity-mode 2>&1 | grep -a '"metric"' > $O/bench_sdxl_fp8.json; cut -c1-260 $O/bench_sdxl_fp8.json
