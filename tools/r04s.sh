#!/bin/bash
timeout 900 python -m pytest tests/test_training_gpu.py -q -x -k "bf16_tensors_of or attention_backward or bf16x1 or graphed or groupnorm or adamw or deterministic" 2>&1 | grep -a "passed\|failed\|Error\|error\|assert" | tail -12
timeout 900 python bench.py --mode train --precision bf16x1 --steps 6 --warmup 2 2>&1 | grep -a '"metric"\|Error\|error' | cut -c1-330
timeout 600 python tools/bench_wgrad.py 2>&1 | tail -12
