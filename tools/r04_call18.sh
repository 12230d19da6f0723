TAG=${1:-r04s}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for m in eager sync; do
echo "== $m" | tee -a $O/probe4.txt
SWN_WINO_MINC=32 timeout 300 python tools/r04_pipe_probe3.py $m 2>&1 | grep -E "^trial|eager2|Error" | tee -a $O/probe4.txt
done
