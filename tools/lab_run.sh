# one gpurun call: ring_lab timings + an SQ counter pass (usage: bash tools/lab_run.sh <tag>)
TAG=${1:-lab1}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 ./tools/ring_lab 10 > $O/ring_lab.txt 2>&1
tail -40 $O/ring_lab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc -o lab -- $R/tools/ring_lab 1 > $O/pmc.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/pmc ${TAG}_sq --out $O > /dev/null 2>&1
rm -rf $O/pmc
ls $O
