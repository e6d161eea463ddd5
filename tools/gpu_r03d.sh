set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests/test_ops_gpu.py tests/test_split_gpu.py -q -x -k "folded or fused_qkv or range_guard" > $O/t_ops.log 2>&1; echo "rc=$?" >> $O/t_ops.log
python tools/bench_fold.py > $O/bench_fold_micro.log 2>&1
python tools/exp_group.py > $O/exp_group.log 2>&1
python -m pytest tests/test_training_gpu.py -q -x -s -k "config3 or bench_train_under_torchrun or deterministic" > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log
python -m pytest tests/test_pipeline_gpu.py -q -x -k "tiny_pipeline_per_step" > $O/t_pipe.log 2>&1; echo "rc=$?" >> $O/t_pipe.log
true
