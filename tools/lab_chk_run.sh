# one gpurun call: ring_lab timings (+ SQ pass), a GPU test subset and the C2 bench line
# usage: bash tools/lab_chk_run.sh <tag> "<pytest files>"
TAG=${1:-labchk}
SEL=${2:-tests/test_ops.py}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
bash tools/lab_run.sh $TAG > $O/lab.log 2>&1
SWAPNET_TEST_VERBOSE=1 timeout 900 python -m pytest $SEL -m gpu -q > $O/tests_gpu.log 2>&1
tail -3 $O/tests_gpu.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python -c "import json;d=json.load(open('$O/bench_c2.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_warp -o warp -- $B --steps 3 --warmup 1 > $O/prof_warp.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/prof_warp ${TAG}_prof_warp --out $O > /dev/null 2>&1
rm -rf $O/prof_warp
grep -E "filter|precut" $O/rocprof_${TAG}_prof_warp_kernel_stats.md
