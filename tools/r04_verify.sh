# round 4: texture-side and fill-dependent GPU tests on the final build (first-layer ring padding of the texture stage, kernel fill)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04verify
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python -m pytest tests/test_pattern_replay.py tests/test_train_parity.py -k "texture" -m gpu -q --durations=8 > $O/t_texture.log 2>&1; echo "texture rc $?" | tee -a $O/rc.txt
tail -14 $O/t_texture.log
timeout 400 python -m pytest tests/test_gradient_penalty.py tests/test_data_parallel.py tests/test_channel_options.py -m gpu -q > $O/t_misc.log 2>&1; echo "misc rc $?" | tee -a $O/rc.txt
tail -4 $O/t_misc.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/rc.txt
echo "wall $(( $(date +%s) - T0 )) s" | tee -a $O/rc.txt
