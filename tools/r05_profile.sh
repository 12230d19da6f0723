# warp C2 on the current build: kernel-trace stats + HBM traffic per kernel (usage: bash tools/r05_profile.sh <tag> [texture])
TAG=${1:-r05a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_warp -o warp -- $B --steps 3 --warmup 1 > $O/prof_warp.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o warp -- $B --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o warp -- $B --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
if [ "$2" = "texture" ]; then
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tex -o tex -- $B --stage texture --steps 3 --warmup 1 > $O/prof_tex.log 2>&1
fi
cd $R
for d in prof_warp pmc_fetch pmc_write prof_tex; do
  [ -d $O/$d ] || continue
  python profiles/summarize_rocprof.py $O/$d ${TAG}_$d --out $O > /dev/null 2>&1
  rm -rf $O/$d
done
python profiles/summarize_rocprof.py traffic ${TAG}_pmc_fetch ${TAG}_pmc_write ${TAG} --out $O
ls $O
