# torch-free A/B of routing switches on the C2 step (usage: bash tools/r05_ab.sh <tag> "SWN_A=1" "SWN_B=2 SWN_C=3" ...)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 300 tools/_bin/native_ab 32 256 20 3 ab "$@" > $O/ab.txt 2>&1
grep "ab mean\|ab round\|losses" $O/ab.txt
if [ -n "$PROF_ENV" ]; then env $PROF_ENV timeout 120 tools/_bin/native_ab 32 256 10 0 prof > $O/prof.txt 2>&1; grep "^prof" $O/prof.txt; fi
