#!/bin/bash
timeout 1500 python -m pytest tests/test_training_gpu.py -q -x 2>&1 | grep -a "passed\|failed\|Error\|error\|assert" | tail -8
timeout 900 python bench.py --mode train --precision bf16x1 --steps 6 --warmup 2 2>&1 | grep -a '"metric"\|Error\|error' | cut -c1-330
