TAG=${1:-r04r}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SWN_WINO_MINC=32 timeout 300 python tools/r04_pipe_probe3.py 2>&1 | grep -E "^trial|Error" | tee -a $O/probe3.txt
timeout 400 python -m pytest tests/test_texture_step.py -m gpu -q 2>&1 | tail -4 | tee -a $O/probe3.txt
