# round 4: one weight-amax pass per layer (forward + input-gradient operands share it): step tests, bench, launch counts
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04verify3
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_warp_step.py tests/test_texture_step.py tests/test_pattern_replay.py -k "not full_resolution and not c2 and not c3 and not one_plane and not unpinned" -m gpu -q > $O/t.log 2>&1; echo "tests rc $?" | tee -a $O/rc.txt
tail -3 $O/t.log
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['value'])"
cd /tmp && export TMPDIR=/tmp
SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o warp -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/prof.log 2>&1
cd $R
python profiles/summarize_rocprof.py $O/prof r04v3 --out $O > /dev/null 2>&1; rm -rf $O/prof
grep -E "amax_partials|conv_precut_kernel|winog_filter_pc" $O/rocprof_r04v3_kernel_stats.md
