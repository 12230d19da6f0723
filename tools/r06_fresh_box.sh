#!/bin/bash
# round 6: the first HIP process of a call against later ones (tools/native_ab ... trace prints the arena hashes after initialisation
# and after every step), then the truth mode (three ways through the step from the same state)  -> gpurun_out/r06_alloc_fill.txt
out=gpurun_out/r06_alloc_fill.txt; : > $out
run() { echo "== $1" >> $out; shift; timeout 120 "$@" 2>&1 | grep -E "^(trace|truth|hash|poisoned|step)" >> $out; }
for p in 1 2 3; do run "process $p" tools/_bin/native_ab 32 256 2 0 trace; done
run "truth" tools/_bin/native_ab 32 256 2 0 truth
cat $out
