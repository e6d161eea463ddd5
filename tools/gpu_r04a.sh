set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 1200 python -m pytest tests/test_training_gpu.py -q -x -k "bf16" > $O/t_bf16.log 2>&1; echo "rc=$?" >> $O/t_bf16.log
timeout 600 python bench.py --mode train --precision bf16x1 --steps 6 --warmup 2 > $O/bench_train_bf16x1.log 2>&1
MFHIP_NO_FLASH_BWD=1 timeout 600 python bench.py --mode train --precision bf16x1 --steps 6 --warmup 2 > $O/bench_train_bf16x1_unfused.log 2>&1
true
