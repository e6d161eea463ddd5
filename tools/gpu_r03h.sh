set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
python -m pytest tests/test_xl_gpu.py tests/test_models_gpu.py -q -x -k "tiny" > $O/t_tiny.log 2>&1; echo "rc=$?" >> $O/t_tiny.log
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_stats -o b -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-parity-mode --no-profile > $O/bench_under_rocprof.log 2>&1 )
cp $(find gpurun_out/r03_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/stats_top.py gpurun_out/r03_stats 40 402 > $O/stats_top.txt 2>&1
rm -rf gpurun_out/r03_stats
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; cp gpurun_out/pmc_gemm_family.json $O/
rm -rf gpurun_out/pmc_step_FETCH_SIZE gpurun_out/pmc_step_WRITE_SIZE
bash tools/pmc_attn.sh > $O/pmc_attn_d40.txt 2>&1; rm -rf gpurun_out/pmca_*
python tools/bench_bw.py > $O/hbm_bandwidth.md 2>&1
python bench.py --mode train --steps 3 --warmup 1 > $O/bench_train.log 2>&1
python tools/gemm_shapes.py > $O/gemm_shapes.txt 2>&1
true
