# per-kernel profiles are taken with the library's second stream off (SWN_OVERLAP=0): with it on, kernels of
# the two streams share the GPU and their individual durations / counters are not attributable
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $R/gpurun_out/tfull.log 2>&1
python bench.py > $R/gpurun_out/bench_c2.json 2> $R/gpurun_out/bench_c2.err
python bench.py --stage texture > $R/gpurun_out/bench_c3.json 2> $R/gpurun_out/bench_c3.err
cd /tmp && export TMPDIR=/tmp
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_warp -o warp -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_warp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_warp_ov -o warp -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_warp_ov.log 2>&1
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tex -o tex -- python $R/bench.py --stage texture --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_tex.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o warp -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_fetch.log 2>&1
SWN_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o warp -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_write.log 2>&1
tail -3 $R/gpurun_out/tfull.log
