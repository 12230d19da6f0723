TAG=${1:-r04t}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in "SWN_NONE=1" "SWN_WINO_MINC=32" "SWN_WINO_MINC=32 SWN_AMAX_FUSED=0" "SWN_WINO_MINC=32 SWN_PAIR=0" "SWN_WINO_MINC=32 SWN_FIRST_RING=0" "SWN_WINO_MINC=32 SWN_PC_PLANES=3" "SWN_WINO_MINC=32 SWN_WINO_S2=0" "SWN_WINO_MINC=32 SWN_PRECUT=0"; do
echo "== $v" | tee -a $O/probe5.txt
env $v timeout 200 python tools/r04_pipe_probe3.py eager 2>&1 | grep -E "^trial|Error" | cut -c1-120 | tee -a $O/probe5.txt
done
