# round 4, GPU call 12: AdamW streamed behind each bucket of the generator's backward pass: equivalence tests + same-box A/B
TAG=${1:-r04l}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_captured_step.py -m gpu -q -x > $O/t_capt.log 2>&1; echo "captured rc $?" | tee -a $O/rc.txt
tail -5 $O/t_capt.log
B="python bench.py --no-roofline --steps 20 --warmup 4"
for v in "" "SWN_STREAM_ADAMW=0" "" "SWN_STREAM_ADAMW=0"; do
  echo "== warp $v" >> $O/ab.txt
  env $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $O/ab.txt
done
for v in "" "SWN_STREAM_ADAMW=0"; do
  echo "== texture $v" >> $O/ab.txt
  env $v $B --stage texture 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $O/ab.txt
done
cat $O/ab.txt
