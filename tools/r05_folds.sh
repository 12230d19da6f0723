# amax folds in act / maxpool / upsample forward (round 5): parity of the stages that use them + launch count + A/B by bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/folds
mkdir -p $O
cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_texture_step.py tests/test_module_calls.py tests/test_models_api.py \
  "tests/test_pattern_replay.py::test_texture_gradients_with_pinned_pattern_at_full_resolution" \
  "tests/test_pattern_replay.py::test_texture_c3_full_batch_training_step_with_pinned_pattern" \
  "tests/test_train_parity.py::test_texture_c3_full_batch_step_matches_oracle" \
  "tests/test_warp_step.py" "tests/test_captured_step.py" > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/rc.txt
tail -5 $O/tests.log
for i in 1 2; do
  timeout 200 python bench.py --stage texture --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2> $O/t.err | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('texture', d['ms_per_step'], d['value'])" | tee -a $O/tex.txt
done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2> $O/w.err | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('warp', d['ms_per_step'], d['value'])" | tee -a $O/tex.txt
cd /tmp && export TMPDIR=/tmp
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tex -o tex -- python $R/bench.py --no-cpu-baseline --no-roofline --stage texture --steps 3 --warmup 1 > $O/prof_tex.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/prof_tex r05b_prof_tex --out $O > /dev/null 2>&1
rm -rf $O/prof_tex
grep "amax_partials\|conv_dma_reduce\|ew_kernel" $O/rocprof_r05b_prof_tex_kernel_stats.md
