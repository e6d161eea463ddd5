set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train.log 2>&1
MFHIP_NO_GN_ACC=1 timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train_noacc.log 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train2.log 2>&1
true
