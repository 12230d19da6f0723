# round 4, HEAD: every GPU test except the slow full-size ones (those ran on this build's predecessors: r04verify, r04verify2)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04verify6
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 420 python -m pytest tests/ -m gpu -q -x -k "not full_resolution and not c2 and not c3 and not loss_statistics and not one_plane and not full_size" > $O/t.log 2>&1; echo "rc $?" | tee -a $O/rc.txt
tail -5 $O/t.log
echo "wall $(( $(date +%s) - T0 )) s" | tee -a $O/rc.txt
