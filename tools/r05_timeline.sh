# two-stream timeline of the C2 step: rocprofv3 kernel trace with the side stream ON, condensed by tools/timeline.py
TAG=${1:-tl}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env $2 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o warp -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 2 > $O/trace.log 2>&1
cd $R
python tools/timeline.py $O/trace > $O/timeline.txt 2>&1
cat $O/timeline.txt
cp $O/trace/*kernel_trace.csv $O/kernel_trace.csv 2>/dev/null; gzip -f $O/kernel_trace.csv; rm -rf $O/trace
