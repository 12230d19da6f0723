# one gpurun call: is the split ring loop power-bound?  Same variants on random and on zero-filled operands, with the shader
# clock each one sustained (s_memtime / s_memrealtime inside a few workgroups).  usage: bash tools/lab_clk_run.sh <tag>
TAG=${1:-labclk}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SET=${2:-3,4,5,7}
LAB_ONLY=$SET timeout 300 ./tools/ring_lab 10 > $O/ring_lab_random.txt 2>&1
LAB_ONLY=$SET LAB_ZERO=1 timeout 300 ./tools/ring_lab 10 > $O/ring_lab_zero.txt 2>&1
grep -A5 "^== exact" $O/ring_lab_random.txt | head -12
grep -A5 "^== exact" $O/ring_lab_zero.txt | head -12
