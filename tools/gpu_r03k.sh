set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
python -m pytest tests/test_split_gpu.py tests/test_ops_gpu.py -q -x -k "split or conv3x3_tiles" > $O/t_split.log 2>&1; echo "rc=$?" >> $O/t_split.log
python bench.py --precision f16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_f16x3_before.log 2>&1
cp ~/.cache/mfhip/tune_cache.json $O/tune_before.json 2>/dev/null
rm -f ~/.cache/mfhip/tune_cache.json
MFHIP_RETUNE=1 MFHIP_TUNE_GRAPH=1 python bench.py --precision f16x3 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_f16x3_retune.log 2>&1
cp ~/.cache/mfhip/tune_cache.json $O/tune_f16x3.json
python bench.py --precision f16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_f16x3_after.log 2>&1
python -m pytest tests/test_pipeline_gpu.py -q -x -s -k "config1_all_50 and f16x3 or config0 and f16x3" > $O/t_pipe.log 2>&1; echo "rc=$?" >> $O/t_pipe.log
true
