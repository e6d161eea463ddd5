set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
for i in 1 2; do
timeout 600 python bench.py --mode train --steps 8 --warmup 2 > $O/bench_train_$i.log 2>&1
MFHIP_NO_FUSED_GRAD_ADD=1 timeout 600 python bench.py --mode train --steps 8 --warmup 2 > $O/bench_train_nofuse_$i.log 2>&1
done
true
