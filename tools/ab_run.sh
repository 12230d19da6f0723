# same-box A/B of refresh / pre-cut switches (ms per step, C2)
TAG=${1:-ab}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for V in "X=0" "SWN_PRECUT=0" "SWN_WINO_PC=0" "SWN_PREFETCH=0" "SWN_PREFETCH=0 SWN_PRECUT=0" "SWN_PREFETCH=0 SWN_WINO_PC=0" "X=1"; do
  env $V python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$V', d['ms_per_step'], d['value'])" >> $O/ab.txt
done
cat $O/ab.txt
