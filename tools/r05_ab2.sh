R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab3
mkdir -p $O
cd $R
timeout 300 tools/_bin/native_ab 32 256 20 3 ab "SWN_TAIL_SPLIT=2" "SWN_TAIL_SPLIT=0" > $O/ab.txt 2>&1
grep "ab mean\|losses" $O/ab.txt
for V in "X=1" "SWN_TAIL_SPLIT=2" "X=2" "SWN_TAIL_SPLIT=2"; do
  env $V timeout 200 python bench.py --stage texture --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2> $O/t.err | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('texture $V', d['ms_per_step'], d['value'])" | tee -a $O/tex.txt
done
