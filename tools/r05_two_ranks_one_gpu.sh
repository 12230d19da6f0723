# N > 1 launch path of bench.py on a ONE-GPU box: two ranks on device 0, gloo process group (RCCL refuses two ranks on one GPU).
# The library-owned exchange is the default: its ncclCommInitRank must FAIL here (duplicate GPU) on both ranks, the ranks must agree
# on it and fall back to the torch.distributed all-reduce -- functional check of parallel.open_native_comm on real hardware.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tworanks
mkdir -p $O
cd $R
export SWAPNET_DIST_BACKEND=gloo SWAPNET_FORCE_DEVICE=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-roofline > $O/out.txt 2> $O/err.txt; echo "rc $?" | tee $O/rc.txt
tail -c 1500 $O/out.txt; echo; grep -i "library-owned\|exchange\|error\|duplicate" $O/err.txt | tail -8
