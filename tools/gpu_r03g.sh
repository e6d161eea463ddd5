set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
rm -f ~/.cache/mfhip/tune_cache.json
MFHIP_TUNE_GRAPH=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_tune.log 2>&1
cp ~/.cache/mfhip/tune_cache.json $O/tune_user_sd15.json
MFHIP_TUNE_GRAPH=1 python bench.py --model sdxl --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_tune_xl.log 2>&1
MFHIP_TUNE_GRAPH=1 python bench.py --model sdxl --precision fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_tune_xl8.log 2>&1
cp ~/.cache/mfhip/tune_cache.json $O/tune_user_all.json
python bench.py --steps 3 --warmup 1 > $O/bench_final.log 2>&1
python -m pytest tests/test_xl_gpu.py tests/test_frontend_gpu.py -q -x > $O/t_xl_front.log 2>&1; echo "rc=$?" >> $O/t_xl_front.log
true
