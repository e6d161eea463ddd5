set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > $O/bench_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 bench.py --mode train --steps 2 --warmup 1 > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_trace*" -delete
true
