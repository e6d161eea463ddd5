set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
bash tools/pmc_probe.sh > $O/pmc_probe.log 2>&1
cp gpurun_out/pmc_probe.txt $O/ 2>/dev/null
rm -rf gpurun_out/pmcp_* 
true
