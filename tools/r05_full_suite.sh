# round 5: the complete GPU suite as the driver runs it (one process, default environment), with durations, then smoke()
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05suite
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=40 > $O/suite.log 2>&1; echo "suite rc $? in $(( $(date +%s) - T0 )) s" | tee -a $O/rc.txt
tail -60 $O/suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/rc.txt
