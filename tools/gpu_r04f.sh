set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests/test_split_gpu.py tests/test_training_gpu.py tests/test_pipeline_gpu.py -q -x -k "f16x3 or split or attention or wgrad or graphed" > $O/t_f16.log 2>&1; echo "rc=$?" >> $O/t_f16.log
timeout 300 python tools/bench_wgrad.py > $O/bench_wgrad.log 2>&1
timeout 300 python tools/bench_split_w.py > $O/bench_split_w.log 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 3 > $O/bench_train.log 2>&1
true
