#!/bin/bash
# round 6: the bias gradient of a layer that forms no input gradient (PatchGAN model.0 in backward_D) on the main stream, beside its weight
# gradient (default), against behind it on the second stream (SWN_BIAS_MAIN=0): bit-identity, ms/step alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_bias; mkdir -p $O; out=$O/ab.txt; : > $out
cd $R
for v in X=1 SWN_BIAS_MAIN=0; do echo "== $v" >> $out; env $v timeout 100 tools/_bin/native_ab 32 256 2 0 hash 2>&1 | grep -E "^hash after" >> $out; done
echo "== truth (default)" >> $out; timeout 100 tools/_bin/native_ab 32 256 2 0 truth 2>&1 | grep -E "^truth (phases|fused|  )" >> $out
for rep in 1 2 3 4 5; do for v in SWN_BIAS_MAIN=0 X=1; do echo -n "$v  " >> $out; env $v timeout 100 tools/_bin/native_ab 32 256 40 0 bench 2>&1 | grep -E "^bench [0-9]" >> $out; done; done
cat $out
