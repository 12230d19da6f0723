#!/bin/bash
# round 6: the cross-entropy term taken early (behind the discriminator's backward pass) against the old order (SWN_CE_EARLY=0):
# bit-identity (hash of two steps, truth mode = phased / two streams / fused), ms/step of 40 steps per process, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ce; mkdir -p $O; out=$O/ab.txt; : > $out
cd $R
for v in X=1 SWN_CE_EARLY=0; do echo "== $v" >> $out; env $v timeout 100 tools/_bin/native_ab 32 256 2 0 hash 2>&1 | grep -E "^hash" >> $out; done
echo "== truth (default)" >> $out; timeout 100 tools/_bin/native_ab 32 256 2 0 truth 2>&1 | grep -E "^truth" >> $out
for rep in 1 2 3 4; do
  for v in SWN_CE_EARLY=0 X=1; do
    echo -n "$v  " >> $out; env $v timeout 100 tools/_bin/native_ab 32 256 40 0 bench 2>&1 | grep -E "^bench [0-9]" >> $out
  done
done
cat $out
