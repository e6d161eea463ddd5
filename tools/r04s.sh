#!/bin/bash
timeout 900 python -m pytest tests/test_training_gpu.py -q -x -k "lds_dma or wgrad or groupnorm_backward" 2>&1 | grep -a "passed\|failed\|Error\|error\|assert" | tail -8
timeout 600 python tools/bench_wgrad.py 2>&1 | tail -11
timeout 900 python bench.py --mode train --precision bf16x1 --steps 10 --warmup 3 2>&1 | grep -a '"metric"\|Error\|error' | cut -c1-330
