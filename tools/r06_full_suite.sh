# round 6: the complete GPU suite as the driver runs it (one process, default environment), with durations, then smoke()
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06suite
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=40 > $O/suite.log 2>&1; echo "suite rc $? in $(( $(date +%s) - T0 )) s" | tee -a $O/rc.txt
tail -60 $O/suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/rc.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" | tee -a $O/rc.txt
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06suite/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r.get('frac_of_sustained'), r['step_hbm_bytes'], r['step_hbm_frac'], d['cpu_baseline']['value'])
PY
