# One short GPU call: every kernel-routing switch of DESIGN.md section 4 against the default, on the C2 step, through the torch-free
# driver (tools/native_ab.cpp, mode `bench`: one process per configuration -- most switches are read when a model is planned, so
# a fresh process per setting is the form that is always right; a process is up in a second).  Each configuration is bracketed by
# default runs (boxes drift by a few 0.1 ms over a minute).   gpurun --timeout 300 -- 'bash tools/r05_sweep.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05sweep
mkdir -p $O
cd $R
run() { env $1 timeout 60 tools/_bin/native_ab 32 256 ${STEPS:-30} 0 bench 2>&1 | grep '^bench ' | awk -v c="$1" '{printf "%-44s %s ms/step %s img/s\n", c, $2, $4}'; }
{
  run "X=default"
  for CFG in "SWN_PC_STAGES=3" "SWN_PC_MI=2" "SWN_TILE256=0" "SWN_WGRAD256=0" "SWN_TILE192=0" "SWN_WINO_VW=4" "SWN_STREAM_ADAMW=0" \
             "SWN_SHARE_DY=0" "SWN_PAIR=0" "SWN_AMAX_FUSED=0" "SWN_FIRST_RING=0" "SWN_WGRAD_PLANES=3" "SWN_PC_PLANES=3" "SWN_PREFETCH=0" \
             "SWN_OVERLAP=0" "SWN_WINO_S2=0" "SWN_TAIL_WINO=0" "SWN_FUSED_IN=0" "SWN_WINO_ADJOINT=0" "SWN_HEAD_TAPN=0" "SWN_WINO_MINC=128" "SWN_WINO_MINC=64" "SWN_WINO_MINC=32" \
             "SWN_PC_PLANES=1 SWN_WGRAD_PLANES=1" ${EXTRA_CFGS}; do
    run "$CFG"
    run "X=default"
  done
} | tee $O/sweep.txt
