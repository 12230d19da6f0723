# round 4, GPU call 1: operator tests + determinism + pinned-pattern parity under default routing, then same-box A/B of
# this round's switches and a kernel-trace profile.   gpurun --timeout 1500 -- 'bash tools/r04_call1.sh'
TAG=${1:-r04a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export SWAPNET_TEST_VERBOSE=1
timeout 500 python -m pytest tests/test_ops.py -m gpu -x -q -s > $O/t_ops.log 2>&1; echo "ops rc $?" | tee -a $O/rc.txt
tail -5 $O/t_ops.log
timeout 400 python -m pytest "tests/test_warp_step.py::test_warp_full_size_properties" "tests/test_pattern_replay.py::test_warp_gradients_with_pinned_pattern_at_full_resolution" -x -q -s > $O/t_props.log 2>&1; echo "props rc $?" | tee -a $O/rc.txt
tail -5 $O/t_props.log
timeout 600 python -m pytest "tests/test_pattern_replay.py::test_warp_c2_full_batch_training_step_with_pinned_pattern" -x -q -s > $O/t_c2.log 2>&1; echo "c2 pinned rc $?" | tee -a $O/rc.txt
tail -5 $O/t_c2.log
for V in X=default SWN_WGRAD_PLANES=3 SWN_AMAX_FUSED=0 SWN_SHARE_DY=0 SWN_PC_STAGES=3 SWN_PC_STAGES=4 X=default2; do
  env $V timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$V', d['ms_per_step'], d['value'])" >> $O/ab_switches.txt
done
cat $O/ab_switches.txt
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python -c "import json;d=json.load(open('$O/bench_c2.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac']);print(json.dumps(d['roofline']['all_gemm_kernels']))"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
SWN_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_warp -o warp -- $B --steps 3 --warmup 1 > $O/prof_warp.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/prof_warp ${TAG}_prof_warp --out $O > /dev/null 2>&1
rm -rf $O/prof_warp
ls $O
