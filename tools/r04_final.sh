#!/bin/bash
# Round-4 closing measurements on one MI355X: every line lands in gpurun_out/<tag>/ (copied to profiles/r04_* afterwards).
O=$MF_SESSION_OUT
R=$GRAFT_REPO_ROOT
cd $R
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== default bench"; timeout 900 python bench.py 2>&1 | grep -a '"metric"' > $O/bench_default.json; cut -c1-400 $O/bench_default.json
echo "== default bench, 3 more"; for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline 2>&1 | grep -a '"metric"' | cut -c1-200; done
echo "== parity mode"; timeout 900 python bench.py --precision f16x3 --no-cpu-baseline 2>&1 | grep -a '"metric"' > $O/bench_f16x3.json; cut -c1-300 $O/bench_f16x3.json
echo "== train bf16x1"; timeout 900 python bench.py --mode train --precision bf16x1 --steps 10 --warmup 3 2>&1 | grep -a '"metric"' > $O/bench_train_bf16x1.json; cut -c1-330 $O/bench_train_bf16x1.json
echo "== train f16x3"; timeout 900 python bench.py --mode train --precision f16x3 --steps 6 --warmup 2 2>&1 | grep -a '"metric"' > $O/bench_train_f16x3.json; cut -c1-330 $O/bench_train_f16x3.json
echo "== train fp32"; timeout 900 python bench.py --mode train --precision fp32 --steps 4 --warmup 2 2>&1 | grep -a '"metric"' > $O/bench_train_fp32.json; cut -c1-330 $O/bench_train_fp32.json
echo "== train bf16x1 + base unet"; timeout 900 python bench.py --mode train --precision bf16x1 --train-base-unet --steps 6 --warmup 2 2>&1 | grep -a '"metric"' > $O/bench_train_bf16x1_unet.json; cut -c1-330 $O/bench_train_bf16x1_unet.json
echo "== wgrad vs forward"; timeout 600 python tools/bench_wgrad.py 2>&1 | grep -a "GFLOP" > $O/wgrad_vs_forward.txt; cat $O/wgrad_vs_forward.txt
cd /tmp && export TMPDIR=/tmp
echo "== rocprof default bench"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o b -- python3 $R/bench.py --no-cpu-baseline 2>&1 | grep -a '"metric"' > $O/bench_under_rocprof.json; cut -c1-300 $O/bench_under_rocprof.json
echo "== rocprof train bench"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python3 $R/bench.py --mode train --precision bf16x1 --steps 10 --warmup 3 2>&1 | grep -a '"metric"' > $O/bench_train_under_rocprof.json; cut -c1-300 $O/bench_train_under_rocprof.json
find $O -name "*kernel_trace.csv" -delete
cd $R
echo "== pmc"; timeout 1500 bash tools/pmc_step.sh 2>&1 | tail -5; cp gpurun_out/pmc_gemm_family.json $O/ 2>/dev/null
