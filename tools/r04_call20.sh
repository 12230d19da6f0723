TAG=${1:-r04u}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for v in "SWN_NONE=1" "SWN_WINO_MINC=32"; do
echo "== $v" | tee -a $O/probe6.txt
env $v timeout 200 python tools/r04_pipe_probe3.py eager 2>&1 | grep -E "^trial|Error" | cut -c1-120 | tee -a $O/probe6.txt
done
