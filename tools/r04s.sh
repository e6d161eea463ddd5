#!/bin/bash
timeout 900 python -m pytest tests/test_training_gpu.py -q -x -k "attention or bf16x1 or bf16_tensors or graphed" 2>&1 | grep -a "passed\|failed\|Error\|error\|assert" | tail -6
timeout 900 python bench.py --mode train --precision bf16x1 --steps 10 --warmup 3 2>&1 | grep -a '"metric"\|Error\|error' | cut -c1-330
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $MF_SESSION_OUT/prof -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --precision bf16x1 --steps 6 --warmup 2 2>&1 | grep '"metric"' | cut -c1-200
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["MF_SESSION_OUT"]+"/prof/**/*kernel_stats.csv",recursive=True)
for p in f:
    rows=list(csv.DictReader(open(p)))
    for r in rows:
        if "attn" in r["Name"]: print(r["Name"][:100], r["Calls"], r["AverageNs"])
PY
find $MF_SESSION_OUT -name "*kernel_trace.csv" -delete
