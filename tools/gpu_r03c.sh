set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "groupnorm" > $O/t_gn.log 2>&1; echo "rc=$?" >> $O/t_gn.log
python -m pytest tests/test_pipeline_gpu.py tests/test_models_gpu.py -q -x -s -k "tiny_pipeline_per_step or config1_batch4 or variants or sd15_single_step or tiny_unet or tiny_brushnet" > $O/t_pipe.log 2>&1; echo "rc=$?" >> $O/t_pipe.log
python tools/bench_fold.py > $O/bench_fold_micro.log 2>&1
python tools/exp_group.py > $O/exp_group.log 2>&1
python tools/bench_gn.py > $O/bench_gn_new.log 2>&1
MFHIP_GN_FUSE_SMALL=1 python tools/bench_gn.py > $O/bench_gn_old.log 2>&1
MFHIP_NO_LNFOLD=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_nofold.log 2>&1
MFHIP_NO_LNFOLD=1 MFHIP_NO_ZC_FOLD=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_nofold_nozc.log 2>&1
MFHIP_NO_LNFOLD=1 MFHIP_GN_FUSE_SMALL=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/bench_nofold_gnold.log 2>&1
MFHIP_NO_LNFOLD=1 python tools/gemm_shapes.py > $O/shapes_nofold.log 2>&1
python tools/gemm_shapes.py > $O/shapes_fold.log 2>&1
true
