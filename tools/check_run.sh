# intermediate GPU check (one gpurun call): a test subset, the C2 bench line and a kernel-trace profile
# usage: bash tools/check_run.sh <tag> "<pytest args>"
TAG=${1:-chk}
SEL=${2:-tests/test_ops.py tests/test_warp_step.py tests/test_texture_step.py}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
KEXPR=${3:-}
if [ -n "$KEXPR" ]; then
  SWAPNET_TEST_VERBOSE=1 timeout 1500 python -m pytest $SEL -m gpu -q -s -k "$KEXPR" > $O/tests_gpu.log 2>&1
else
  SWAPNET_TEST_VERBOSE=1 timeout 1500 python -m pytest $SEL -m gpu -q -s > $O/tests_gpu.log 2>&1
fi
grep -E 'native .* torch fp32|passed|failed|FAILED|one-signed' $O/tests_gpu.log | tail -80
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python -c "import json;d=json.load(open('$O/bench_c2.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'])"
SWN_PRECUT=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_noprecut.json 2> /dev/null
SWN_TAIL_WINO=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_nofusedin.json 2> /dev/null
python -c "import json;print('A/B same box: no-precut', json.load(open('$O/bench_c2_noprecut.json'))['ms_per_step'], ' no-tail-winograd', json.load(open('$O/bench_c2_nofusedin.json'))['ms_per_step'])"
timeout 300 python bench.py --stage texture --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python -c "import json;d=json.load(open('$O/bench_c3.json'));print(d['value'],d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
SWN_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_warp -o warp -- $B --steps 3 --warmup 1 > $O/prof_warp.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/prof_warp ${TAG}_prof_warp --out $O > /dev/null 2>&1
rm -rf $O/prof_warp
head -45 $O/rocprof_${TAG}_prof_warp_kernel_stats.md
