# N > 1 launch path of bench.py on a ONE-GPU box, round 6: two ranks on device 0, gloo process group (RCCL refuses two ranks on one GPU).
# (a) the default since round 6: the torch.distributed all-reduce per bucket (the library-owned exchange is opt-in for world > 1);
# (b) the bf16 wire format; (c) SWAPNET_NATIVE_COMM=1: ncclCommInitRank must FAIL on both ranks (duplicate GPU), the ranks agree stage by
# stage and fall back; (d) bench.py launching its own ranks must refuse: one device visible, --gpus 2.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tworanks6
mkdir -p $O
cd $R
export SWAPNET_DIST_BACKEND=gloo SWAPNET_FORCE_DEVICE=0
B="bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-roofline"
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
timeout 300 $L --master-port 29533 $B > $O/a.out 2> $O/a.err; echo "a rc $?" | tee $O/rc.txt
timeout 300 $L --master-port 29534 $B --grad-wire bf16 > $O/b.out 2> $O/b.err; echo "b rc $?" | tee -a $O/rc.txt
SWAPNET_NATIVE_COMM=1 timeout 300 $L --master-port 29535 $B > $O/c.out 2> $O/c.err; echo "c rc $?" | tee -a $O/rc.txt
python bench.py --gpus 2 --steps 1 > $O/d.out 2> $O/d.err; echo "d rc $? (must be non-zero)" | tee -a $O/rc.txt
for f in a b c; do echo "== $f"; tail -1 $O/$f.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','n_gpus','world','exchange','exchange_wire','exchange_bytes_per_step','losses_finite','dist_backend')})"; grep -i "library-owned\|ncclCommInitRank" $O/$f.err | tail -3; done
echo "== d"; tail -2 $O/d.err
