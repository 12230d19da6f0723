# round 4, GPU call 6: pair-form Winograd planes
TAG=${1:-r04f}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export SWAPNET_TEST_VERBOSE=1
timeout 500 python -m pytest tests/test_ops.py -m gpu -x -q -k "conv_forward or conv_backward or split_main_loop or heavy" > $O/t_ops.log 2>&1; echo "ops rc $?" | tee -a $O/rc.txt
tail -3 $O/t_ops.log
timeout 600 python -m pytest "tests/test_pattern_replay.py::test_warp_gradients_with_pinned_pattern_at_full_resolution" "tests/test_pattern_replay.py::test_texture_gradients_with_pinned_pattern_at_full_resolution" "tests/test_warp_step.py::test_warp_full_size_properties" -x -q -s > $O/t_props.log 2>&1; echo "props rc $?" | tee -a $O/rc.txt
grep -n "flips\|passed\|failed" $O/t_props.log | tail -8
for V in X=default SWN_PAIR=0 X=default2 SWN_PAIR=0; do
  env $V timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-roofline 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$V', d['ms_per_step'], d['value'])" >> $O/ab_switches.txt
done
cat $O/ab_switches.txt
SWN_PROF_DETAIL=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null > $O/bench_detail.json
python - <<EOF
import json
d=json.load(open("$O/bench_detail.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"])
for k,v in sorted(d["roofline"]["all_gemm_kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:24]: print("%-95s %7.1f TF/s %7.3f ms" % (k, v["tflops"], v["ms_per_step"]))
EOF
ls $O
