set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train.log 2>&1
MFHIP_NO_PRESPLIT_MULTI=1 timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train_nomulti.log 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train2.log 2>&1
true
