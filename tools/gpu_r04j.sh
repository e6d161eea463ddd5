set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python bench.py --mode train --precision fp32 --steps 3 --warmup 3 > $O/bench_train_fp32.log 2>&1
timeout 900 python bench.py --mode train --train-base-unet --steps 4 --warmup 3 > $O/bench_train_base_unet.log 2>&1
true
