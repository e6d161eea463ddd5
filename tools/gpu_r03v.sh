set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1
true
