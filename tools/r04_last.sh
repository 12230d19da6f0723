R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 --with-h2d > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --stage texture --steps 12 --warmup 4 > $O/bench_c3.json 2> $O/bench_c3.err
SWAPNET_BENCH_RCCL1=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_rccl_world1.json 2> $O/bench_c2_rccl_world1.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
wc -l $O/bench_c2.json $O/bench_c3.json $O/bench_c2_rccl_world1.json $O/bench_default.json
tail -c 300 $O/bench_c2_rccl_world1.json
