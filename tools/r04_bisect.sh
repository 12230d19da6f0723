# round 4: the two GPU tests that failed in the full-suite run, under one switch at a time
TAG=${1:-r04bis}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export SWAPNET_TEST_KEEP_SWITCHES=1
T="tests/test_models_api.py::test_two_stage_device_pipeline_equals_reference_two_pass_inference tests/test_warp_step.py::test_warp_two_steps_match_oracle_and_reference"
for v in "" "SWN_AMAX_FUSED=0" "SWN_PAIR=0" "SWN_WINO_VW=4" "SWN_FIRST_RING=0" "SWN_SHARE_DY=0" "SWN_WGRAD_PLANES=3" "SWN_PC_PLANES=3" "SWN_OVERLAP=0" "SWN_PREFETCH=0" "SWN_WINO_PC=0" "SWN_WINO_S2=0"; do
  echo "== [$v]" | tee -a $O/bisect.txt
  env $v timeout 300 python -m pytest $T -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|AssertionError: \(" | cut -c1-400 | tee -a $O/bisect.txt
done
