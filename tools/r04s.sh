#!/bin/bash
timeout 600 python -m pytest tests/test_training_gpu.py -q -x -k "attention_backward_bf16" -s 2>&1 | grep "attention backward\|passed\|failed"
timeout 2400 python -m pytest tests/test_training_gpu.py -q -x 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $MF_SESSION_OUT/prof -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --precision bf16x1 --steps 6 --warmup 2 2>&1 | tail -2
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["MF_SESSION_OUT"]+"/prof/**/*kernel_stats.csv",recursive=True)
for p in f:
    rows=list(csv.DictReader(open(p)))
    for r in rows[:40]:
        print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
