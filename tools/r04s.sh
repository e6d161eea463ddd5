#!/bin/bash
timeout 600 python -m pytest tests/test_training_gpu.py -q -x -k "attention_backward_bf16" -s 2>&1 | grep "attention backward\|passed\|failed\|Error"
timeout 900 python -m pytest tests/test_training_gpu.py -q -x -k "bf16x1" 2>&1 | tail -5
timeout 900 python bench.py --mode train --precision bf16x1 --steps 6 --warmup 2 2>&1 | grep '"metric"'
