# one gpurun call: the texture pinned-pattern test under a set of switches (which change moved its error?)
TAG=${1:-bis}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
T="tests/test_pattern_replay.py::test_texture_gradients_with_pinned_pattern_at_full_resolution"
for V in "default" "SWN_WINO_ADJOINT=2" "SWN_WINO_S2=0" "SWN_FUSED_IN=0" "SWN_PRECUT=0"; do
  echo "== $V" >> $O/bisect.log
  if [ "$V" = "default" ]; then
    SWAPNET_TEST_VERBOSE=1 timeout 600 python -m pytest "$T" -m gpu -q -s -k eval 2>&1 | grep -E "flips|rel-L2|passed|failed" >> $O/bisect.log
  else
    env $V SWAPNET_TEST_VERBOSE=1 timeout 600 python -m pytest "$T" -m gpu -q -s -k eval 2>&1 | grep -E "flips|rel-L2|passed|failed" >> $O/bisect.log
  fi
done
cat $O/bisect.log
