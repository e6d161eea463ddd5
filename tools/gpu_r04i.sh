set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_stats -o b -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-parity-mode --no-profile > $O/bench_under_rocprof.log 2>&1 )
cp $(find gpurun_out/r04_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf gpurun_out/r04_stats
python bench.py --no-cpu-baseline --no-parity-mode > $O/bench_same_box.log 2>&1
true
