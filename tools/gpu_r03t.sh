set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
timeout 600 python -m pytest tests/test_training_gpu.py -q -s -k "attention_backward" > $O/t_attn_bwd.log 2>&1; echo "rc=$?" >> $O/t_attn_bwd.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > $O/bench_train.log 2>&1
true
