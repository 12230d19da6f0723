#!/bin/bash
cd /root/repo; O=gpurun_out
tools/_bin/graph_replay_cost 10 > $O/r06_graph_cost.txt 2>&1
SWAPNET_TEST_KEEP_SWITCHES=1 python -m pytest tests/test_pattern_replay.py tests/test_reference_goldens_full_res.py tests/test_texture_step.py tests/test_train_parity.py -x -q -m gpu -s --durations=8 2>&1 | grep -E "worst|flips|passed|failed|rel-L2|^[0-9.]+s " > $O/r06_verify1.txt
