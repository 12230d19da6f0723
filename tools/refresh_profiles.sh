# Produces everything under profiles/ for one round: run on the GPU box as
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r05'
# then copy the condensed files from gpurun_out/<tag>/ into profiles/ (README there lists the names).
# Per-kernel profiles are taken with the library's second stream off (SWN_OVERLAP=0): with it on, kernels of the two
# streams share the GPU and their individual durations / counters are not attributable.  Counter passes (--pmc) are
# separate runs without any trace domain, one step each (bench.py --no-roofline runs exactly the timed steps).
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 --with-h2d > $O/bench_c2.json 2> $O/bench_c2.err
python -c "import json;d=json.load(open('$O/bench_c2.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'],d['roofline']['pipe_util_nominal_step'],d.get('cpu_baseline'))"
python bench.py --stage texture --steps 12 --warmup 4 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --precision f16 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_c2_f16.json 2> $O/bench_c2_f16.err
python bench.py --captured --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_c2_captured.json 2> $O/bench_c2_captured.err
python bench.py --stage infer > $O/bench_infer.json 2> $O/bench_infer.err
python bench.py --stage joint --steps 12 --warmup 4 > $O/bench_joint.json 2> $O/bench_joint.err
python bench.py --stage joint --captured --steps 12 --warmup 4 > $O/bench_joint_captured.json 2> $O/bench_joint_captured.err
SWAPNET_BENCH_RCCL1=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_rccl_world1.json 2> $O/bench_c2_rccl_world1.err
# same-box A/B of this round's switches (ms/step)
for V in X=default SWN_CONV_STATS=0 SWN_TAIL_SPLIT=0 SWN_PAIR=0 SWN_AMAX_FUSED=0 SWN_PREFETCH=0 SWN_STREAM_ADAMW=0 SWN_OVERLAP=0 X=default2; do
  env $V python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2> /dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$V', d['ms_per_step'], d['value'])" >> $O/ab_switches.txt
done
cat $O/ab_switches.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_warp -o warp -- $B --steps 3 --warmup 1 > $O/prof_warp.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o warp -- $B --steps 1 --warmup 0 > $O/pmc_sq.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/pmc_sq2 -o warp -- $B --steps 1 --warmup 0 > $O/pmc_sq2.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o warp -- $B --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o warp -- $B --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tex -o tex -- $B --stage texture --steps 3 --warmup 1 > $O/prof_tex.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_tex -o tex -- $B --stage texture --steps 1 --warmup 0 > $O/pmc_fetch_tex.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_tex -o tex -- $B --stage texture --steps 1 --warmup 0 > $O/pmc_write_tex.log 2>&1
# keep the merge small: the raw per-dispatch CSVs are condensed on the box
cd $R
for d in prof_warp prof_tex pmc_fetch pmc_write pmc_sq pmc_sq2 pmc_fetch_tex pmc_write_tex; do
  python profiles/summarize_rocprof.py $O/$d ${TAG}_$d --out $O > /dev/null 2>&1
  rm -rf $O/$d
done
python profiles/summarize_rocprof.py traffic ${TAG}_pmc_fetch ${TAG}_pmc_write ${TAG} --out $O
python profiles/summarize_rocprof.py traffic ${TAG}_pmc_fetch_tex ${TAG}_pmc_write_tex ${TAG}_texture --out $O
ls $O
# round 6 extras: what a recorded hipGraph costs on this runtime and how its replay of the step overlaps (tools/graph_replay_cost.hip,
# tools/r06_timeline.sh), the ragged last round of the 36-plane launch and the remedies measured (tools/tile_lab.hip variants 24 ..)
tools/_bin/graph_replay_cost 10 > $O/graph_replay_cost.txt 2>&1
bash tools/r06_timeline.sh > /dev/null 2>&1; cp $R/gpurun_out/r06_tl/timeline_eager.txt $O/timeline_eager.txt; cp $R/gpurun_out/r06_tl/timeline_captured.txt $O/timeline_captured.txt
(for sh in 0 1; do LAB_SHAPE=$sh LAB_ONLY=0,1,2,3,4,24,25,26,27,28,29,30 timeout 300 tools/tile_lab 10; done) > $O/tile_lab_ragged.txt 2>&1
tools/_bin/native_ab 32 256 10 2 phases > $O/native_phases.txt 2>&1
ls $O
