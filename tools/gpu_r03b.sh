set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "folded or fused_qkv" > $O/t_ops.log 2>&1; echo "rc=$?" >> $O/t_ops.log
python -m pytest tests/test_pipeline_gpu.py -q -x -s -k "config1_all_50 and bf16 or config0 and bf16 or config1_batch4 and bf16" > $O/t_pipe.log 2>&1; echo "rc=$?" >> $O/t_pipe.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_fold.log 2>&1
MFHIP_NO_LNFOLD=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_nofold.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_fold2.log 2>&1
cp ~/.cache/mfhip/tune_cache.json $O/tune_user.json 2>/dev/null
true
