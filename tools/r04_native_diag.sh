# The very last GPU seconds of round 4: where do swn_model_step and swn_model_step_dp part (tools/native_ab.cpp, mode "diag")?
O=$GRAFT_REPO_ROOT/gpurun_out/r04x
mkdir -p $O
cd $GRAFT_REPO_ROOT
NATIVE_AB_SKIP_OPS=1 timeout ${1:-18} tools/_bin/native_ab 32 256 1 0 diag > $O/native_diag.txt 2>&1
echo "rc $?" >> $O/native_diag.txt
tail -45 $O/native_diag.txt
