set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "conv3x3_tiles or warp_specialised or folded or fused_qkv or attention" > $O/t_ops.log 2>&1; echo "rc=$?" >> $O/t_ops.log
for q in 4 3 2 0; do echo "Q64=$q"; MFHIP_ATTN_Q64=$q python tools/bench_attn.py 2>&1 | grep -v amdgpu; done > $O/bench_attn.log 2>&1
python -m pytest tests/test_frontend_gpu.py -q -x > $O/t_front.log 2>&1; echo "rc=$?" >> $O/t_front.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_a.log 2>&1
MFHIP_ATTN_Q64=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_noq64.log 2>&1
MFHIP_NO_LNFOLD=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_nofold.log 2>&1
MFHIP_RETUNE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_retune.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_b.log 2>&1
cp ~/.cache/mfhip/tune_cache.json $O/tune_user.json 2>/dev/null
python -m pytest tests/test_xl_gpu.py -q -x -s -k "config4" > $O/t_xl.log 2>&1; echo "rc=$?" >> $O/t_xl.log
true
