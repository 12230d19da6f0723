# (HISTORY: the first GPU call of round 5 as prepared by round 4; the 64-row-tile kernel and the SWAPNET_UNVERIFIED_GPU gate it names are gone since.)
# (Round 4, third session: items 3 / 4 below are ANSWERED -- tools/native_ab.cpp ran the 64-row wave tiles (bit-equal, +0.59 ms/step:
#  stays off) and the library-owned exchange with RCCL at world 1 from plain C++ (profiles/native_ab_r04.txt, native_diag_r04.txt).
#  What is left for this call: the three opt-in PYTHON tests, smoke(), the default bench line, the native exchange through
#  parallel.NativeComm.  For switch A/Bs use the torch-free driver: a C2 model is up in a second --
#      tools/_bin/native_ab 32 256 20 3 ab "SWN_X=1" "SWN_Y=2 SWN_Z=0"       (alternating blocks in one process)
#      env SWN_ONCE=1 tools/_bin/native_ab 32 256 40 0 bench                  (switches a process reads once)
#  build: hipcc -O2 -std=c++17 tools/native_ab.cpp -Iinclude -Lswapnet_amd/csrc -lswapnet_hip -ldl -Wl,-rpath,'$ORIGIN/../../swapnet_amd/csrc' -o tools/_bin/native_ab)
# Round 5, FIRST GPU call (≈ 12 GPU-minutes): everything round 4 wrote after its GPU budget was spent, before anything else is built on it.
#   gpurun --timeout 1100 -- 'bash tools/r05_first_call.sh'
# 1. the opt-in GPU tests (SWAPNET_UNVERIFIED_GPU=1): 256 x 128 tile of 64-row wave tiles (conv_fwd_pcm_kernel) against the shipped
#    128 x 128 kernel; gradient penalty at PatchGAN depths 2 / 4; the library-owned exchange with real RCCL at world size 1
# 2. smoke() + the default bench line (the product path did not change: same ISA for every shipped kernel, tools/isa_diff.py)
# 3. same-box A/B, ms/step of C2: default | SWN_PC_MI=2 | native exchange at world 1 (SWAPNET_BENCH_RCCL1=1 with / without
#    SWAPNET_NATIVE_COMM=1)
# 4. kernel-trace of the SWN_PC_MI=2 run: conv_fwd_pcm_kernel vs conv_fwd_pc_kernel<4,4,2,4,2,true> average launch time
# If (1) is green: drop the skipif marks of those tests; if SWN_PC_MI=2 wins in (3): make it the default for launches with
# >= pcm_min_tiles() tiles (conv_gemm.hip conv_fwd) and re-record the routing digest.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05first
mkdir -p $O
cd $R
export SWAPNET_UNVERIFIED_GPU=1
timeout 420 python -m pytest -m gpu -q -x \
  "tests/test_ops.py::test_64_row_wave_tiles_match_the_128x128_ring_kernel" \
  "tests/test_data_parallel.py::test_one_rank_native_rccl_exchange_equals_the_fused_step" \
  "tests/test_gradient_penalty.py::test_gradient_penalty_at_other_patchgan_depths" \
  "tests/test_joint_step.py" \
  "tests/test_pixel_discriminator.py" \
  "tests/test_ops.py::test_wavefront_gather_roi_align_is_bit_identical" \
  "tests/test_ops.py::test_winograd_layers_of_129_to_192_channels" \
  -s > $O/t_unverified.log 2>&1; echo "unverified-tests rc $?" | tee -a $O/rc.txt
tail -15 $O/t_unverified.log
unset SWAPNET_UNVERIFIED_GPU
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/rc.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" | tee -a $O/rc.txt
tail -c 600 $O/bench_default.json
timeout 200 python bench.py --stage joint --steps 10 --warmup 3 > $O/bench_joint.json 2> $O/bench_joint.err; echo "bench joint rc $?" | tee -a $O/rc.txt
tail -c 400 $O/bench_joint.json
timeout 60 tools/_bin/native_ab 32 256 10 0 prof > $O/native_prof.txt 2>&1; tail -30 $O/native_prof.txt
timeout 200 python bench.py --stage joint --captured --steps 10 --warmup 3 > $O/bench_joint_cap.json 2> $O/bench_joint_cap.err; echo "bench joint captured rc $?" | tee -a $O/rc.txt
tail -c 400 $O/bench_joint_cap.json
for V in "X=0" "SWAPNET_BENCH_RCCL1=1" "SWAPNET_BENCH_RCCL1=1 SWAPNET_NATIVE_COMM=1"; do
  env $V timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2> $O/ab.err | tail -1 | \
    python -c "import sys,json;d=json.loads(sys.stdin.read());print('$V', d['ms_per_step'], d['value'], d.get('exchange'))" >> $O/ab.txt
done
cat $O/ab.txt
