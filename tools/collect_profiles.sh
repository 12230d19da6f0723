#!/bin/bash
# Copy what tools/refresh_profiles.sh left under gpurun_out/<tag>/ into the tracked profiles/ directory under the
# names DESIGN.md, profiles/README.md and bench.py refer to.   usage: tools/collect_profiles.sh r04
set -e
TAG=${1:-r06}
S=gpurun_out/$TAG
D=profiles
cp $S/bench_c2.json                              $D/bench_${TAG}_c2.json
cp $S/bench_c3.json                              $D/bench_${TAG}_c3_texture.json
cp $S/bench_c2_f16.json                          $D/bench_${TAG}_c2_f16.json
cp $S/bench_c2_captured.json                     $D/bench_${TAG}_c2_captured.json
cp $S/bench_infer.json                           $D/bench_${TAG}_infer.json
cp $S/bench_joint.json                           $D/bench_${TAG}_joint.json
cp $S/bench_joint_captured.json                  $D/bench_${TAG}_joint_captured.json
cp $S/bench_c2_rccl_world1.json                  $D/bench_${TAG}_c2_rccl_world1.json
cp $S/ab_switches.txt                            $D/ab_switches_${TAG}.txt
cp $S/rocprof_${TAG}_prof_warp_kernel_stats.md   $D/rocprof_${TAG}_warp_c2_kernel_stats.md
cp $S/rocprof_${TAG}_prof_tex_kernel_stats.md    $D/rocprof_${TAG}_texture_c3_kernel_stats.md
for c in sq sq2 fetch write; do cp $S/pmc_${TAG}_pmc_$c.json $D/pmc_${TAG}_$c.json; done
for c in fetch write; do cp $S/pmc_${TAG}_pmc_${c}_tex.json $D/pmc_${TAG}_${c}_texture.json; done
cp $S/traffic_${TAG}.json $S/traffic_${TAG}_texture.json $D/
for f in graph_replay_cost timeline_eager timeline_captured tile_lab_ragged native_phases; do [ -f $S/$f.txt ] && cp $S/$f.txt $D/${f}_${TAG}.txt; done
[ -f $D/tile_lab_ragged_${TAG}.txt ] && mv $D/tile_lab_ragged_${TAG}.txt $D/tile_lab_${TAG}_ragged.txt
ls -la $D | grep "_${TAG}[_.]"
