# one gpurun call for the two-plane (fp16 x 2) pre-cut kernel: the op-level tests, the pinned-pattern gradient comparison and the
# warp step test, then the C2 bench line in both plane forms on the same box, and a kernel trace.
# usage: bash tools/h2_run.sh <tag> ["<pytest files>"]
TAG=${1:-h2}
SEL=${2:-tests/test_ops.py tests/test_pattern_replay.py tests/test_warp_step.py}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SWAPNET_TEST_VERBOSE=1 timeout 1200 python -m pytest $SEL -m gpu -q -s -x > $O/tests_gpu.log 2>&1
grep -E 'native .* torch fp32|passed|failed|FAILED|Error|error|one-signed|pinned|max rel' $O/tests_gpu.log | tail -60
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python -c "import json;d=json.load(open('$O/bench_c2.json'));print('planes 2:', d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['frac'])" || tail -5 $O/bench_c2.err
SWN_PC_PLANES=3 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2_planes3.json 2> $O/bench_c2_planes3.err
python -c "import json;d=json.load(open('$O/bench_c2_planes3.json'));print('planes 3:', d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['frac'])" || tail -5 $O/bench_c2_planes3.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
SWN_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_warp -o warp -- $B --steps 3 --warmup 1 > $O/prof_warp.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/prof_warp ${TAG}_prof_warp --out $O > /dev/null 2>&1
rm -rf $O/prof_warp
head -16 $O/rocprof_${TAG}_prof_warp_kernel_stats.md
grep -E "amax|precut|filter_pc" $O/rocprof_${TAG}_prof_warp_kernel_stats.md
