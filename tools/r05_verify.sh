# Round 5: the GPU tests touched by this round's changes (pruned kernels, un-split tails, wave RoIAlign default, asynchronous
# parameter block of the captured step, library-owned exchange default) + the bench lines that go with them.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05verify
mkdir -p $O
cd $R
timeout 900 python -m pytest -m gpu -q --durations=15 \
  tests/test_boundary.py tests/test_ops.py tests/test_captured_step.py tests/test_data_parallel.py \
  "tests/test_joint_step.py::test_c5_shaped_joint_step_of_both_gan_pairs" \
  "tests/test_train_parity.py::test_warp_c2_full_batch_step_matches_oracle" \
  "tests/test_train_parity.py::test_texture_c3_full_batch_step_matches_oracle" \
  "tests/test_train_parity.py::test_warp_c2_full_batch_step_with_winograd_forms_on_every_level" \
  > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/rc.txt
tail -30 $O/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/rc.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" | tee -a $O/rc.txt
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05verify/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r.get('frac_of_sustained'), r.get('sustained'))
PY
for A in "--captured" "--stage joint" "--stage joint --captured" "--stage texture"; do
  timeout 300 python bench.py $A --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2> $O/b.err | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$A', d['ms_per_step'], d['value'])" | tee -a $O/lines.txt
done
SWAPNET_BENCH_RCCL1=1 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2> $O/rccl1.err | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('rccl1', d['ms_per_step'], d['value'], d.get('exchange'))" | tee -a $O/lines.txt
SWAPNET_BENCH_RCCL1=1 SWAPNET_NATIVE_COMM=0 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2> $O/rccl1t.err | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('rccl1 torch', d['ms_per_step'], d['value'], d.get('exchange'))" | tee -a $O/lines.txt
