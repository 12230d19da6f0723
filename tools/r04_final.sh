# round 4, last GPU call: the tests that failed in the full-suite run (now fixed), then the profile refresh of the final build
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
timeout 500 python -m pytest "tests/test_models_api.py" "tests/test_warp_step.py::test_warp_two_steps_match_oracle_and_reference" tests/test_captured_step.py "tests/test_gradient_penalty.py::test_library_rng_keeps_the_layout_pad_channels_out_of_the_penalty" -m gpu -q > $O/t_fixed.log 2>&1; echo "fixed-tests rc $?" | tee -a $O/rc.txt
tail -4 $O/t_fixed.log
bash tools/refresh_profiles.sh r04
