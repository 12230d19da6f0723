R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/phase4
mkdir -p $O
cd $R
timeout 600 python -m pytest -m gpu -q -x tests/test_ops.py tests/test_texture_step.py tests/test_warp_step.py "tests/test_train_parity.py::test_texture_c3_full_batch_step_matches_oracle" > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/rc.txt
tail -8 $O/tests.log
timeout 300 tools/_bin/native_ab 32 256 20 3 ab "SWN_PHASE4=0" > $O/ab.txt 2>&1
grep "ab mean" $O/ab.txt
for V in "X=1" "SWN_PHASE4=0" "X=2" "SWN_PHASE4=0"; do
  env $V timeout 200 python bench.py --stage texture --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2> $O/t.err | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('texture $V', d['ms_per_step'], d['value'])" | tee -a $O/tex.txt
done
