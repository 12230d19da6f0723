# after the eleven A/B switches were folded to their defaults (device code unchanged by tools/isa_diff.py): smoke, the routing-asserting
# full-size tests, the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05last
mkdir -p $O
cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee $O/rc.txt
timeout 600 python -m pytest -m gpu -q -x "tests/test_train_parity.py::test_warp_c2_full_batch_step_matches_oracle" "tests/test_train_parity.py::test_texture_c3_full_batch_step_matches_oracle" tests/test_captured_step.py tests/test_boundary.py > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/rc.txt
tail -3 $O/tests.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/rc.txt
python -c "
import json,os
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['routing'])"
