TAG=${1:-r04q}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export SWN_WINO_MINC=32
for m in none create run graph run; do
  timeout 200 python tools/r04_pipe_probe2.py $m 2>&1 | grep -E "^mode|^eager2|^warp model|Error" | tee -a $O/probe2.txt
done
