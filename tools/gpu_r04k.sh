set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04k_stats -o b -- python3 bench.py --model sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-profile > $O/bench_sdxl_under_rocprof.log 2>&1 )
cp $(find gpurun_out/r04k_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_sdxl.csv
rm -rf gpurun_out/r04k_stats
true
