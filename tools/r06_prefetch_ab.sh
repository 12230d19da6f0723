#!/bin/bash
# round 6: the operand refresh issued piecewise (default) against everything at the top of the pass (SWN_PREFETCH=3), lookahead sweep;
# bit-identity (hash of two steps), ms/step of 40 steps per process, alternating; then the timeline of the default
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_prefetch; mkdir -p $O; out=$O/ab.txt; : > $out
cd $R
for v in X=1 SWN_PREFETCH=3; do echo "== $v" >> $out; env $v timeout 100 tools/_bin/native_ab 32 256 2 0 hash 2>&1 | grep -E "^hash" >> $out; done
for rep in 1 2 3; do
  for v in SWN_PREFETCH=3 X=1 SWN_PREFETCH_AHEAD=2 SWN_PREFETCH_AHEAD=8 SWN_PREFETCH_AHEAD=16 SWN_PREFETCH=0; do
    echo -n "$v  " >> $out; env $v timeout 100 tools/_bin/native_ab 32 256 40 0 bench 2>&1 | grep -E "^bench [0-9]" >> $out
  done
done
cat $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- $R/tools/_bin/native_ab 32 256 6 0 bench > $O/tl.log 2>&1
cd $R && python tools/timeline.py $O/tl --dump-gap > $O/timeline_piecewise.txt 2>&1; rm -rf $O/tl
grep -n "^gap\|step window\|busy:" $O/timeline_piecewise.txt | cut -c1-250
