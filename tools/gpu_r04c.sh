set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 bench.py --mode train --precision bf16x1 --steps 2 --warmup 1 > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_trace*" -delete
true
