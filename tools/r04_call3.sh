# round 4, GPU call 3: one-plane configuration, per-shape GEMM table, texture line, PMC passes of the warp step
TAG=${1:-r04c}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest "tests/test_pattern_replay.py::test_one_plane_configuration_tolerance_study" -x -q -s > $O/t_f16.log 2>&1; echo "f16 study rc $?" | tee -a $O/rc.txt
grep -n "one fp16 plane\|passed\|failed\|Error" $O/t_f16.log | tail -5
SWN_PROF_DETAIL=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_c2_detail.json 2> $O/bench_c2_detail.err
python - <<EOF
import json
d=json.load(open("$O/bench_c2_detail.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["pipe_util_nominal"], d["roofline"]["pipe_util_nominal_step"], d["config"]["routing"])
for k,v in sorted(d["roofline"]["all_gemm_kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:45]: print("%-95s %7.1f TF/s %7.3f ms" % (k, v["tflops"], v["ms_per_step"]))
EOF
timeout 300 python bench.py --steps 12 --warmup 4 --precision f16 --no-cpu-baseline > $O/bench_c2_f16.json 2> $O/bench_c2_f16.err
python -c "import json;d=json.load(open('$O/bench_c2_f16.json'));print('f16', d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['frac'],d['losses_finite'])"
timeout 400 python bench.py --stage texture --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err
python -c "import json;d=json.load(open('$O/bench_c3.json'));print('texture', d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['frac'],d.get('cpu_baseline'))"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
SWN_OVERLAP=0 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o warp -- $B --steps 1 --warmup 0 > $O/pmc_sq.log 2>&1
SWN_OVERLAP=0 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/pmc_sq2 -o warp -- $B --steps 1 --warmup 0 > $O/pmc_sq2.log 2>&1
SWN_OVERLAP=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o warp -- $B --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
SWN_OVERLAP=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o warp -- $B --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
SWN_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tex -o tex -- $B --stage texture --steps 3 --warmup 1 > $O/prof_tex.log 2>&1
cd $R
for d in prof_tex pmc_fetch pmc_write pmc_sq pmc_sq2; do
  python profiles/summarize_rocprof.py $O/$d ${TAG}_$d --out $O > /dev/null 2>&1
  rm -rf $O/$d
done
python profiles/summarize_rocprof.py traffic ${TAG}_pmc_fetch ${TAG}_pmc_write ${TAG} --out $O
ls $O
