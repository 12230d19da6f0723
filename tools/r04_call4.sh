# round 4, GPU call 4: captured training step, one-plane tolerance study
TAG=${1:-r04d}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 700 python -m pytest tests/test_captured_step.py -m gpu -x -q -s > $O/t_capt.log 2>&1; echo "captured rc $?" | tee -a $O/rc.txt
tail -4 $O/t_capt.log
timeout 600 python -m pytest "tests/test_pattern_replay.py::test_one_plane_configuration_tolerance_study" -x -q -s > $O/t_f16.log 2>&1; echo "f16 study rc $?" | tee -a $O/rc.txt
grep -n "one fp16 plane\|passed\|failed\|Error" $O/t_f16.log | tail -5
for V in "" "--captured" "" "--captured"; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $V 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('[$V]', d['ms_per_step'], d['value'], d['config']['step_form'], d['losses_finite'])" >> $O/ab_captured.txt
done
timeout 300 python bench.py --stage texture --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --captured 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('texture captured', d['ms_per_step'], d['value'])" >> $O/ab_captured.txt
cat $O/ab_captured.txt
ls $O
