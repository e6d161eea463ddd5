set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode --no-profile > $O/bench_default.log 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 600 python bench.py --no-cpu-baseline --no-parity-mode --no-profile > $O/bench_devkernarg.log 2>&1
GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --no-cpu-baseline --no-parity-mode --no-profile > $O/bench_hwq8.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode --no-profile > $O/bench_default2.log 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 600 python bench.py --mode train --steps 6 --warmup 3 > $O/bench_train_devkernarg.log 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 3 > $O/bench_train_default.log 2>&1
true
