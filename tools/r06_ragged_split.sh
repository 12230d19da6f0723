#!/bin/bash
# round 6 (measured, not kept: profiles/ragged_split_r06.txt): a launch whose last round would be ragged issued in pieces against one launch
# (SWN_RAGGED_SPLIT=0; the switch existed only in the experiment's build): bit-identity, per-shape launch times in order on one stream
# (SWN_PROF_DETAIL=1), step time in alternating blocks of one process and in alternating processes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ragged; mkdir -p $O; out=$O/ab.txt; : > $out
cd $R
for v in X=1 SWN_RAGGED_SPLIT=0; do echo "== $v" >> $out; env $v timeout 100 tools/_bin/native_ab 32 256 2 0 hash 2>&1 | grep -E "^hash after" >> $out; done
for v in X=1 SWN_RAGGED_SPLIT=0; do echo "== prof $v" >> $out; env $v SWN_PROF_DETAIL=1 timeout 200 tools/_bin/native_ab 32 256 5 0 prof 2>&1 | grep -E "^prof" | grep -E "conv_fwd_pc|total|steps in order" >> $out; done
echo "== ab (alternating blocks of 20 steps, one process)" >> $out
timeout 300 tools/_bin/native_ab 32 256 20 4 ab "SWN_RAGGED_SPLIT=0" 2>&1 | grep -E "^ab" >> $out
for rep in 1 2 3; do for v in SWN_RAGGED_SPLIT=0 X=1; do echo -n "$v  " >> $out; env $v timeout 100 tools/_bin/native_ab 32 256 40 0 bench 2>&1 | grep -E "^bench [0-9]" >> $out; done; done
cat $out
