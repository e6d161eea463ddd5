#!/bin/bash
timeout 3400 python -m pytest tests -q -x -m gpu 2>&1 | grep -a "passed\|failed\|Error\|error" | tail -8
O=$MF_SESSION_OUT; R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py --mode train --precision bf16x1 --steps 10 --warmup 3 2>&1 | grep -a '"metric"' > $O/bench_train_bf16x1.json; cut -c1-330 $O/bench_train_bf16x1.json
timeout 900 python bench.py --mode train --precision bf16x1 --train-base-unet --steps 6 --warmup 2 2>&1 | grep -a '"metric"' > $O/bench_train_bf16x1_unet.json; cut -c1-330 $O/bench_train_bf16x1_unet.json
timeout 900 python bench.py --mode train --precision f16x3 --steps 6 --warmup 2 2>&1 | grep -a '"metric"' > $O/bench_train_f16x3.json; cut -c1-330 $O/bench_train_f16x3.json
timeout 900 python bench.py --mode train --precision fp32 --steps 4 --warmup 2 2>&1 | grep -a '"metric"' > $O/bench_train_fp32.json; cut -c1-330 $O/bench_train_fp32.json
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python3 $R/bench.py --mode train --precision bf16x1 --steps 10 --warmup 3 2>&1 | grep -a '"metric"' > $O/bench_train_under_rocprof.json; cut -c1-300 $O/bench_train_under_rocprof.json
find $O -name "*kernel_trace.csv" -delete
cd $R
timeout 600 python bench.py 2>&1 | grep -a '"metric"' > $O/bench_default.json; cut -c1-200 $O/bench_default.json
