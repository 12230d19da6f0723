#!/bin/bash
# round 6: HBM bytes of one C2 step in the reduced-precision configuration (bench.py --precision f16: one fp16 plane per operand), same
# counter passes as tools/refresh_profiles.sh  -> gpurun_out/r06_f16/traffic_r06_f16.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_f16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --precision f16"
SWN_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o warp -- $B --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o warp -- $B --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
cd $R
for d in pmc_fetch pmc_write; do python profiles/summarize_rocprof.py $O/$d r06f16_$d --out $O > /dev/null 2>&1; rm -rf $O/$d; done
python profiles/summarize_rocprof.py traffic r06f16_pmc_fetch r06f16_pmc_write r06_f16 --out $O
ls $O
