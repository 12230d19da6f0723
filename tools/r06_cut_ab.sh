#!/bin/bash
# the fused low-plane cut (conv_gemm.hip resid_pack): bit-equality of the instruction sequence, of two training steps (weight-arena hashes
# of the previous library build, LD_PRELOADed, against this one), and the same-box timing of both builds
cd /root/repo; O=gpurun_out/r06_cut.txt
tools/_bin/mixtest > $O 2>&1
echo "== previous build (tools/_bin/old/libswapnet_hip.so)" >> $O
LD_PRELOAD=$PWD/tools/_bin/old/libswapnet_hip.so tools/_bin/native_ab 32 256 2 0 hash 2>&1 | grep -E "^hash" >> $O
echo "== this build" >> $O
tools/_bin/native_ab 32 256 2 0 hash 2>&1 | grep -E "^hash" >> $O
for i in 1 2 3; do
  echo "old: $(LD_PRELOAD=$PWD/tools/_bin/old/libswapnet_hip.so tools/_bin/native_ab 32 256 20 0 bench 2>&1 | grep '^bench [0-9]')" >> $O
  echo "new: $(tools/_bin/native_ab 32 256 20 0 bench 2>&1 | grep '^bench [0-9]')" >> $O
done
SWN_PROF_DETAIL=1 tools/_bin/native_ab 32 256 5 1 prof 2>&1 | grep -E "wgrad_dma|pc_128x128\[|pc_256x64|total" | sort -k7 -n -r | head -14 >> $O
