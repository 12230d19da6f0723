#!/bin/bash
# conv-epilogue InstanceNorm statistics: parity at 256 x 256 against the reference goldens, then same-box timing of the C2 step with and without
cd /root/repo; O=gpurun_out
python -m pytest tests/test_reference_goldens_full_res.py tests/test_warp_step.py -x -q -m gpu -s 2>&1 | tail -15 > $O/r06_stats_tests.txt
for i in 1 2; do
  tools/_bin/native_ab 32 256 10 2 bench 2>&1 | tail -2
  SWN_CONV_STATS=0 tools/_bin/native_ab 32 256 10 2 bench 2>&1 | tail -2
done > $O/r06_stats_ab.txt 2>&1
