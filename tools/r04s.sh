#!/bin/bash
O=$MF_SESSION_OUT; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o b -- python3 $R/bench.py --no-cpu-baseline --no-parity-mode 2>&1 | grep -a '"metric"' > $O/bench_under_rocprof.json; cut -c1-300 $O/bench_under_rocprof.json
find $O -name "*kernel_trace.csv" -delete
cd $R
timeout 900 python bench.py --no-cpu-baseline --no-parity-mode 2>&1 | grep -a '"metric"' > $O/bench_same_box.json; cut -c1-300 $O/bench_same_box.json
