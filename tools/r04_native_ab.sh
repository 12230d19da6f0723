# The last 0.8 GPU-minutes of round 4: no time for `import torch` on a fresh box -- the torch-free C-ABI driver instead
# (tools/native_ab.cpp: op-level bit-equality of the 64-row wave tiles, same-process A/B of the C2 step, RCCL world-1 exchange).
O=$GRAFT_REPO_ROOT/gpurun_out/r04x
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout ${1:-45} tools/_bin/native_ab 32 256 20 3 > $O/native_ab.txt 2>&1
echo "rc $?" >> $O/native_ab.txt
tail -40 $O/native_ab.txt
