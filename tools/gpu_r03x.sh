set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
timeout 600 python -m pytest tests/test_training_gpu.py -q -x -k "wgrad" > $O/t_wgrad.log 2>&1; echo "rc=$?" >> $O/t_wgrad.log
timeout 300 python tools/bench_wgrad.py > $O/bench_wgrad.log 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train.log 2>&1
MFHIP_WGRAD_128=1 timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train128.log 2>&1
true
