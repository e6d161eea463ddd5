set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
python bench.py > $O/bench_default.log 2>&1
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_stats -o b -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-parity-mode --no-profile > $O/bench_under_rocprof.log 2>&1 )
cp $(find gpurun_out/r03_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf gpurun_out/r03_stats
python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
true
