set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
python -m pytest tests/test_ops_gpu.py -q -x -k "attention" > gpurun_out/r03a/t_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r03a/t_attn.log
python tools/bench_attn.py > gpurun_out/r03a/bench_attn.log 2>&1
python -m pytest tests/test_pipeline_gpu.py -q -x -s -k "config1_all_50 or tiny_pipeline_per_step or config0 or config1_batch4" > gpurun_out/r03a/t_pipe.log 2>&1; echo "rc=$?" >> gpurun_out/r03a/t_pipe.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > gpurun_out/r03a/bench_new.log 2>&1
MFHIP_NO_TEMB_TABLE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > gpurun_out/r03a/bench_notab.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > gpurun_out/r03a/bench_new2.log 2>&1
tail -3 gpurun_out/r03a/*.log
