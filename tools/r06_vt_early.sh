#!/bin/bash
# round 6: the texture stage's target VGG16 features taken early (behind the discriminator's backward pass) against the reference's place
# inside backward_G (SWN_VT_EARLY=0): ms/step of bench.py --stage texture alternating, then the texture / joint / module GPU tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_vt; mkdir -p $O; out=$O/ab.txt; : > $out
cd $R
for rep in 1 2 3; do
  for v in SWN_VT_EARLY=0 X=1; do
    env $v python bench.py --stage texture --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$v', d['ms_per_step'], d['value'], d['losses_finite'])" >> $out
  done
done
for v in SWN_VT_EARLY=0 X=1; do env $v python bench.py --stage joint --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('joint $v', d['ms_per_step'], d['value'])" >> $out; done
cat $out
python -m pytest tests/test_texture_step.py tests/test_joint_step.py tests/test_module_calls.py tests/test_reference_goldens_full_res.py tests/test_captured_step.py -x -q -m gpu 2>&1 | tail -4
