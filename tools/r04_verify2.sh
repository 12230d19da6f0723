# round 4: the C2 (warp, bs 32) parity tests and the operator tests on the final build
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04verify2
mkdir -p $O
cd $R
timeout 700 python -m pytest tests/test_train_parity.py tests/test_pattern_replay.py -k "warp_c2" -m gpu -q --durations=6 > $O/t_c2.log 2>&1; echo "c2 rc $?" | tee -a $O/rc.txt
tail -10 $O/t_c2.log
timeout 300 python -m pytest tests/test_ops.py tests/test_split_numerics.py tests/test_module_calls.py -m gpu -q > $O/t_ops.log 2>&1; echo "ops rc $?" | tee -a $O/rc.txt
tail -3 $O/t_ops.log
