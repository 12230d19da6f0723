# round 4: bench.py's multi-rank launch shape on one GPU (two ranks on device 0, gloo): the JSON line must still be the last line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04verify4
mkdir -p $O
cd $R
SWAPNET_DIST_BACKEND=gloo SWAPNET_FORCE_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench2.out 2> $O/bench2.err; echo "rc $?" | tee -a $O/rc.txt
wc -l $O/bench2.out; tail -1 $O/bench2.out | cut -c1-300; tail -3 $O/bench2.err
