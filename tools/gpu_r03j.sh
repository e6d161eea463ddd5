set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "groupnorm" > $O/t_gn.log 2>&1; echo "rc=$?" >> $O/t_gn.log
MFHIP_GN_UNROLL=8 MFHIP_GN_APPLY_BLOCKS=256 MFHIP_GN_CHUNKS=32 python -m pytest tests/test_ops_gpu.py -q -x -k "groupnorm" > $O/t_gn8.log 2>&1; echo "rc=$?" >> $O/t_gn8.log
bash tools/bench_gn_sweep.sh > $O/gn_sweep.log 2>&1
true
