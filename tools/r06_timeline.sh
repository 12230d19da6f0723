#!/bin/bash
# captured (hipGraph replay) vs eager step on a timeline: rocprofv3 --kernel-trace of both, condensed by tools/timeline.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --steps 4 --warmup 3"
rocprofv3 --kernel-trace --output-format csv -d $O/eager -o t -- $B > $O/eager.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/captured -o t -- $B --captured > $O/captured.log 2>&1
cd $R
python tools/timeline.py $O/eager > $O/timeline_eager.txt 2>&1
python tools/timeline.py $O/captured > $O/timeline_captured.txt 2>&1
tail -1 $O/eager.log | cut -c1-200; tail -1 $O/captured.log | cut -c1-200
rm -rf $O/eager $O/captured
