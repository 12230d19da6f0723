# round 4, GPU call 7: coalesced pre-cut producers, narrow PatchGAN input gradient
TAG=${1:-r04g}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_ops.py -m gpu -x -q -k "conv_forward or conv_backward or split_main_loop" > $O/t_ops.log 2>&1; echo "ops rc $?" | tee -a $O/rc.txt
tail -3 $O/t_ops.log
timeout 600 python -m pytest "tests/test_pattern_replay.py::test_warp_gradients_with_pinned_pattern_at_full_resolution" "tests/test_warp_step.py::test_warp_step_at_full_resolution_matches_oracle" "tests/test_captured_step.py" -m gpu -x -q -s > $O/t_props.log 2>&1; echo "props rc $?" | tee -a $O/rc.txt
grep -n "flips\|passed\|failed" $O/t_props.log | tail -8
for V in X=default X=default2; do
  env $V timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$V', d['ms_per_step'], d['value'])" >> $O/ab_switches.txt
done
timeout 300 python bench.py --stage texture --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('texture', d['ms_per_step'], d['value'])" >> $O/ab_switches.txt
timeout 300 python bench.py --precision f16 --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('f16', d['ms_per_step'], d['value'])" >> $O/ab_switches.txt
cat $O/ab_switches.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
SWN_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_warp -o warp -- $B --steps 3 --warmup 1 > $O/prof_warp.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/prof_warp ${TAG}_prof_warp --out $O > /dev/null 2>&1
rm -rf $O/prof_warp
ls $O
