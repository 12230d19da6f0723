# one gpurun call: tools/tile_lab (usage: bash tools/r05_lab.sh <tag> ["ENV=.. ENV=.."])
TAG=${1:-lab}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
env $2 timeout 500 ./tools/tile_lab 10 > $O/tile_lab.txt 2>&1
cat $O/tile_lab.txt
if [ -n "$3" ]; then env $3 timeout 500 ./tools/tile_lab 10 > $O/tile_lab_b.txt 2>&1; cat $O/tile_lab_b.txt; fi
