# round 4, GPU call 10: 2-channel-per-thread Winograd transforms + row-parallel filter transform: ops tests, A/B bench, kernel stats
TAG=${1:-r04j}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -x > $O/t_ops.log 2>&1; echo "ops rc $?" | tee -a $O/rc.txt
tail -5 $O/t_ops.log
B="python bench.py --no-roofline --steps 20 --warmup 4"
for v in "" "SWN_WINO_VW=4" ""; do
  echo "== $v" >> $O/ab.txt
  env $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $O/ab.txt
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o warp -- python $R/bench.py --no-roofline --steps 3 --warmup 1 > $O/prof.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/prof r04j --out $O
rm -rf $O/prof
grep -iE "wino|tailw|conv_fwd_pc|wgrad" $O/rocprof_r04j_kernel_stats.md | cut -c1-150 | head -40
