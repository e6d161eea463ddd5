#!/bin/bash
timeout 3400 python -m pytest tests -q -x -m gpu 2>&1 | grep -a "passed\|failed\|Error\|error" | tail -8
