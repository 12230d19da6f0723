# round 4, GPU call 8: the slow parity tests under the default routing (C2 / C3, full resolution), verbose error tables
TAG=${1:-r04h}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export SWAPNET_TEST_VERBOSE=1
timeout 1100 python -m pytest tests/test_train_parity.py -m gpu -q -s --durations=15 > $O/t_parity.log 2>&1; echo "parity rc $?" | tee -a $O/rc.txt
grep -n "worst rel-L2\|passed\|failed\|FAILED\|s call" $O/t_parity.log | tail -30
timeout 700 python -m pytest "tests/test_texture_step.py" "tests/test_pattern_replay.py::test_texture_c3_full_batch_training_step_with_pinned_pattern" "tests/test_warp_step.py::test_warp_step_at_full_resolution_matches_oracle" -m gpu -q -s --durations=10 > $O/t_tex.log 2>&1; echo "tex rc $?" | tee -a $O/rc.txt
grep -n "flips\|passed\|failed\|FAILED\|s call\|ratio" $O/t_tex.log | tail -40
ls $O
