set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 600 python -m pytest tests/test_training_gpu.py -q -x -s -k "attention_backward" > $O/t_attn_bwd.log 2>&1; echo "rc=$?" >> $O/t_attn_bwd.log
timeout 600 python -m pytest tests/test_split_gpu.py tests/test_ops_gpu.py -q -x -k "attention" > $O/t_attn.log 2>&1; echo "rc=$?" >> $O/t_attn.log
timeout 900 python -m pytest tests/test_training_gpu.py -q -x -k "two_training_steps or deterministic or config3" > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log
timeout 600 python bench.py --mode train --steps 3 --warmup 1 > $O/bench_train_flash.log 2>&1
MFHIP_NO_FLASH_BWD=1 timeout 600 python bench.py --mode train --steps 3 --warmup 1 > $O/bench_train_unfused.log 2>&1
true
