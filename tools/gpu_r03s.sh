set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
timeout 1200 python -m pytest tests/test_training_gpu.py -q -x > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > $O/bench_train.log 2>&1
MFHIP_NO_PRESPLIT=1 timeout 600 python bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_nopresplit.log 2>&1
true
