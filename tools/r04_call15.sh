TAG=${1:-r04p}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in "SWN_WINO_MINC=32" "SWN_WINO_MINC=32 SWN_AMAX_FUSED=0" "SWN_WINO_MINC=32 SWN_PAIR=0" "SWN_WINO_MINC=32 SWN_PC_PLANES=3" "SWN_X=1"; do
  echo "== [$v]" | tee -a $O/probe.txt
  env $v timeout 200 python tools/r04_pipe_probe.py 2>&1 | grep -E "^graph|Error" | tee -a $O/probe.txt
done
SWAPNET_TEST_KEEP_SWITCHES=1 timeout 300 python -m pytest "tests/test_warp_step.py::test_warp_two_steps_match_oracle_and_reference" -m gpu -q -s 2>&1 | grep -E "^step|passed|failed|Error" | cut -c1-400 | tee -a $O/probe.txt
