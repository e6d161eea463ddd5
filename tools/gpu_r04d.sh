set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
export MFHIP_TUNE_CACHE=$PWD/gpurun_out/r04d/user_tune.json
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train_before.log 2>&1
MFHIP_RETUNE=1 MFHIP_TUNE_GRAPH=1 timeout 2400 python bench.py --mode train --steps 1 --warmup 1 > $O/retune.log 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train_after.log 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train_after2.log 2>&1
true
