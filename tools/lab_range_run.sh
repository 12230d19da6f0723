# one gpurun call: accuracy of the two-plane fp16 split (gemm_h, A cut in the loop) against the shipped three-plane bf16 loop when the
# operands have a wide dynamic range or gradient-like magnitudes, with and without the power-of-two operand scale.
# usage: bash tools/lab_range_run.sh <tag>
TAG=${1:-labrange}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() { echo "### $1"; env $1 LAB_ONLY=3,7 timeout 200 ./tools/ring_lab 3 | grep -A4 -E "^s_memrealtime|^== exact" | grep -v "^--"; }
{
run "LAB_EXP_SPREAD=7"
run "LAB_EXP_SPREAD=20"
run "LAB_EXP_SPREAD=30"
run "LAB_A_SHIFT=20"
run "LAB_A_SHIFT=20 LAB_A_KSCALE=20"
run "LAB_A_SHIFT=20 LAB_A_KSCALE=12"
run "LAB_A_SHIFT=20 LAB_A_KSCALE=28"
} > $O/ring_lab_range.txt 2>&1
cat $O/ring_lab_range.txt
