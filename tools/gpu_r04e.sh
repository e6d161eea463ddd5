set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_training_gpu.py -q -x -k "graphed" > $O/t_graph.log 2>&1; echo "rc=$?" >> $O/t_graph.log
timeout 600 python bench.py --mode train --steps 6 --warmup 3 > $O/bench_train_graph.log 2>&1
MF_TRAIN_GRAPH=0 timeout 600 python bench.py --mode train --steps 6 --warmup 3 > $O/bench_train_eager.log 2>&1
timeout 600 python bench.py --mode train --precision bf16x1 --steps 6 --warmup 3 > $O/bench_train_bf16x1_graph.log 2>&1
MF_TRAIN_GRAPH=0 timeout 600 python bench.py --mode train --precision bf16x1 --steps 6 --warmup 3 > $O/bench_train_bf16x1_eager.log 2>&1
true
