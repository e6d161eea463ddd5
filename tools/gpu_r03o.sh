set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout 600 python -m pytest tests/test_training_gpu.py -q -x -s -k "groupnorm_backward or layernorm" > $O/t_norm.log 2>&1; echo "rc=$?" >> $O/t_norm.log
timeout 300 python tools/bench_norm_bwd.py > $O/bench_norm.log 2>&1
timeout 900 python -m pytest tests/test_training_gpu.py -q -x > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > $O/bench_train.log 2>&1
true
