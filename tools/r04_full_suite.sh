# round 4: the complete GPU suite as the driver runs it (one process, default environment), with durations, then smoke()
TAG=${1:-r04full}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu --durations=30 > $O/suite.log 2>&1; echo "suite rc $?" | tee -a $O/rc.txt
echo "suite wall $(( $(date +%s) - T0 )) s" | tee -a $O/rc.txt
tail -45 $O/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/rc.txt
tail -3 $O/smoke.log
