#!/usr/bin/env python
"""bench.py -- images/sec of the full warp-stage G+D step (BaseGAN.optimize_parameters,
models/base_gan.py:194-203 of the reference) at 256x256, bs 32 per GPU, fp32, on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = forward, backward_D, AdamW(D), backward_G, AdamW(G) on one synthetic batch that is
already resident in HBM (config C2 of BASELINE.json: "Warp stage, 256x256 synthetic, bs=32,
1xMI355X, fp32").  Train mode (dropout on), random-init weights of the reference architecture
(kaiming, the reference's default), smooth GAN labels redrawn every step like the reference.
N > 1: pure data parallel, weak scaling (bs 32 per GPU), RCCL all-reduce of the two flat
gradient arenas; the generator's backward runs in buckets (swn_model_backward_G_part) and each
bucket's all-reduce overlaps the back-propagation of the earlier layers.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel family: FLOPs the kernel EXECUTES (2*M*N*K per launch) / HIP-event time of
                  those launches (`achieved`, fp32-equivalent TFLOP/s) vs `peak` = the nominal dense fp16 MFMA peak
                  (2500 TFLOP/s) / the 16-bit MFMA products the kernel issues per fp32 product (3 in the two-plane
                  form) -> `frac`.  `sustained` = what swn_probe_mfma measures in this run for the same instruction
                  mix on register-resident operands (random / zero), with the shader clock read inside the kernel;
                  `frac_of_sustained` = achieved against that.  Whole step, two readings:
                  `step_frac_executed` = executed (fp32-equivalent) FLOPs of all implicit-GEMM launches of one step /
                  step time / the fp32-MFMA peak (157.3) -- kept on that yardstick across rounds;
                  `algorithmic_speedup` = dense-conv FLOPs of the step (251.34 GFLOP/img)
                  / executed FLOPs (Winograd F(4x4,3x3) / F(3x3,4x4) and the folded tail conv execute fewer);
                  `dense_equivalent_frac` = their product (dense-count FLOPs / step time / peak), which exceeds
                  step_frac_executed by exactly that factor and is NOT a utilisation figure.
                  `traffic` = HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
                  passes of this same command (`traffic_source`), not collected inside this run.
  cpu_baseline -- the CPU oracle (a port of the reference step, oracle/swapnet_oracle.py) timed
                  on this box's host cores on a bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_IMG_256 = 251.34          # BASELINE.md section 3 (dense-conv definition)
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md "Peak FP32 (matrix)": v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 (v_mfma_f32_32x32x16_bf16)
# SWN_SPLIT=0 runs every GEMM on v_mfma_f32_32x32x2_f32 (IEEE fp32 products, the 157.3 TFLOP/s pipe); the default forms each fp32
# product on the 16-bit pipe from two fp16 planes per operand (next paragraph).  PEAK_SPLIT_TFLOPS = 2500 / 6 is only the YARDSTICK of
# the round-2 / round-3 lines (their three-plane bf16 form issued 6 MFMAs per product; that form left the library in round 5), kept
# so `frac_of_bf16x6_roofline` can be followed across rounds; the one kernel family that still cuts operands three ways in its loop
# is conv_fwd_dma_* (0.09 ms of the C2 step).
SPLIT = os.environ.get("SWN_SPLIT", "1") != "0"
PEAK_SPLIT_TFLOPS = round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)
# The pre-cut forward-type ring kernel (conv_fwd_pc_*) takes its operands as TWO fp16 planes of (operand x 2^k), k from the
# operand's amax: x = h + l, products h h + h l + l h = 3 fp16 MFMAs per fp32 product (same dense rate as bf16), error of the
# dropped l l term 2^-22.  (The three-plane bf16 form of rounds 2-3 is gone from the library: round 5.)
PC_PLANES = {"1": 1}.get(os.environ.get("SWN_PC_PLANES", "2"), 2)
PEAK_PC_TFLOPS = round(PEAK_BF16_MFMA_TFLOPS / {1: 1.0, 2: 3.0}.get(PC_PLANES, 6.0), 1)
# Round 4: the weight-gradient ring kernel takes the same two-plane form (both operands cut in the loop, scales from amax slots
# their producers fill).
WGRAD_PLANES = {"1": 1}.get(os.environ.get("SWN_WGRAD_PLANES", "2"), 2)


def mfma16_per_product(kernel_name):
    """16-bit MFMA products the kernel family issues per fp32 product it delivers (0 = it multiplies on the f32 MFMA pipe):
    the step's `pipe_util_nominal` prices the EXECUTED matrix work against the nominal 2.5 PFLOP/s, whatever the formulation."""
    if not SPLIT:
        return 0
    if "_pc_" in kernel_name:
        return {1: 1, 2: 3}.get(PC_PLANES, 6)
    if "conv_wgrad_dma" in kernel_name:
        return {1: 1, 2: 3}.get(WGRAD_PLANES, 6)
    if "_dma_" in kernel_name:
        return 6
    return 0


def cpu_baseline(sample_bs=4, size=256, warm=2, timed=5):
    """Oracle = functional port of the reference's WarpModel step on torch CPU ("kind": "port": /root/reference does not
    exist on the GPU box; the port is pinned to golden vectors recorded from the real reference, tests/test_oracle_golden.py)."""
    from oracle import swapnet_oracle as O
    # threads actually usable by this process (cgroup / affinity aware), capped: beyond ~32 threads
    # torch-CPU convolutions at this size stop scaling and oversubscription only slows them down
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(sample_bs, size, size, seed=1234)
    st = O.WarpStepOracle(G, D, training=True)
    for _ in range(warm):
        st.step(*batch)
    t0 = time.time()
    for _ in range(timed):
        st.step(*batch)
    dt = (time.time() - t0) / timed
    out = {"value": round(sample_bs / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": f"warp G+D step {size}x{size} bs {sample_bs}, {warm} warm-up + {timed} timed steps, "
                     f"torch {torch.__version__} CPU fp32, {cores} threads"}
    # BASELINE.json configs[0] (C1): warp 64x64, bs 4 on the CPU path -- one "epoch" of 8 synthetic batches after 2 warm-up steps
    batch1 = O.synth_warp_batch(4, 64, 64, seed=1234)
    st1 = O.WarpStepOracle(G, D, training=True)
    for _ in range(2):
        st1.step(*batch1)
    t0 = time.time()
    for _ in range(8):
        st1.step(*batch1)
    d1 = (time.time() - t0) / 8
    out["c1_64x64_bs4"] = {"value": round(4 / d1, 3), "unit": "images/sec", "s_per_step": round(d1, 4), "steps": 8}
    return out


def cpu_baseline_texture(sample_bs=2, size=256, warm=1, timed=4):
    """The texture-stage step (TextureModel.optimize_parameters, models/texture_model.py:121-180) of the same port on the host
    cores: RoIAlign + pix2pix U-Net + PatchGAN + VGG16 perceptual / style losses, seeded-random VGG weights, train mode."""
    from oracle import swapnet_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    G, D, vgg = O.texture_module_params(img_size=size), O.patchgan_params(22), O.vgg16_feature_params()
    batch = O.synth_texture_batch(sample_bs, size, size, seed=1234)
    st = O.TextureStepOracle(G, D, vgg, training=True)
    for _ in range(warm):
        st.step(*batch)
    t0 = time.time()
    for _ in range(timed):
        st.step(*batch)
    dt = (time.time() - t0) / timed
    return {"value": round(sample_bs / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"texture G+D step {size}x{size} bs {sample_bs} (12 ROIs, L1 + VGG16 content + style), {warm} warm-up + {timed} "
                      f"timed steps, torch {torch.__version__} CPU fp32, {cores} threads"}


def _rccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:          # noqa: BLE001 -- informational field only
        return None


def bench_infer(args):
    """inference.py's two stages (warp -> label hand-off -> texture) at batch size 1, the
    reference's inference batch size (inference.py:67), kept on the device and replayed as a hipGraph.  Reports the
    per-image latency (host launch -> results complete, inputs resident) eager and as a graph replay."""
    from swapnet_amd import engine, synthetic
    from swapnet_amd.modules import init_tensor
    torch.cuda.set_device(0)
    ctx = engine.Context(device=0, workspace_mb=256)
    B, S = 1, args.size
    warp = engine.NativeModel(ctx, "warp", B, S, S, is_train=False)
    tex = engine.NativeModel(ctx, "texture", B, S, S, is_train=False)
    torch.manual_seed(0)
    for m in (warp, tex):
        m.load_state_dict(engine.NET_G, {n: (torch.zeros(sh) if n.endswith(".bias") else init_tensor(torch.empty(sh), "kaiming"))
                                         for n, sh in m.param_infos(engine.NET_G).items()})
    wb, tb = synthetic.warp_batch(B, S, S, seed=1), synthetic.texture_batch(B, S, S, seed=2)
    warp.set_input(0, wb["bodys"]); warp.set_input(1, wb["input_cloths"])
    tex.set_input(0, tb["input_textures"]); tex.set_input(1, tb["rois"])
    pipe = engine.NativePipeline(warp, tex)
    res = {}
    for mode, use_graph in (("eager", False), ("hipgraph", True)):
        for _ in range(max(args.warmup, 2)):
            pipe.run(use_graph)
        torch.cuda.synchronize()
        lat = []
        for _ in range(max(args.steps, 50)):
            t0 = time.perf_counter()
            pipe.run(use_graph)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        lat.sort()
        res[mode] = {"p50_ms": round(lat[len(lat) // 2], 4), "p90_ms": round(lat[int(len(lat) * 0.9)], 4),
                     "min_ms": round(lat[0], 4)}
    out = {"metric": "two-stage inference latency (warp -> texture), 256x256, batch 1", "value": res["hipgraph"]["p50_ms"],
           "unit": "ms/image (p50)", "n_gpus": 1, "steps": max(args.steps, 50), "warmup": max(args.warmup, 2),
           "ms_per_step": res["hipgraph"]["p50_ms"], "higher_is_better": False, "scaling": "n/a", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"inference.py pipeline: WarpModule forward -> argmax labels -> one-hot -> TextureModule "
                                  f"forward (12 ROIs), {S}x{S}, batch 1, eval mode, inputs resident in HBM"},
           "latency": res, "images_per_sec": round(1e3 / res["hipgraph"]["p50_ms"], 2)}
    print(json.dumps(out), flush=True)


def bench_joint(args):
    """BASELINE.json config C5's shape on ONE GPU: both G/D pairs in one process -- the warp stage at DeepFashion's 4:3 (256 x 192)
    and the texture stage at the square crop (its U-Net depth follows a square img_size: SURVEY.md section 5 caveat), bs 16 each,
    training mode, alternating swn_model_step (or swn_model_step_captured with --captured) on one context.  Not the headline
    (that is C2); the parity side of this configuration is tests/test_joint_step.py.  A "step" = one optimize_parameters of EACH
    model; value = images through both stages per second."""
    from swapnet_amd import engine, synthetic
    from swapnet_amd.modules import init_tensor
    torch.cuda.set_device(0)
    ctx = engine.Context(device=0, workspace_mb=1024)
    B = 16 if args.batch == 32 else args.batch
    H, S = args.size, args.size
    W = args.size * 3 // 4 if (args.size * 3 // 4) % 64 == 0 else args.size // 2       # 4:3 at 256 (192); 2:1 where 4:3 is no multiple of 64
    warp = engine.NativeModel(ctx, "warp", B, H, W, is_train=True, dropout=0.5)
    tex = engine.NativeModel(ctx, "texture", B, S, S, is_train=True, dropout=0.5)
    torch.manual_seed(0)
    for m, nets in ((warp, (engine.NET_G, engine.NET_D)), (tex, (engine.NET_G, engine.NET_D, engine.NET_VGG))):
        for net in nets:
            m.load_state_dict(net, {n: (torch.zeros(sh) if n.endswith(".bias") else init_tensor(torch.empty(sh), "kaiming"))
                                    for n, sh in m.param_infos(net).items()})
        m.set_hyper()
    synthetic.fill_inputs(warp, "warp", B, H, W, seed=1234)
    synthetic.fill_inputs(tex, "texture", B, S, S, seed=1235)
    rng = torch.Generator().manual_seed(4321)
    n = [0]

    def one_step():
        n[0] += 1
        for m, salt in ((warp, 0), (tex, 500)):
            lab = [float(torch.rand(1, generator=rng) * 0.4 + 0.7) for _ in range(3)]
            m.step(lab, training=True, seed=n[0] * 1000 + salt, captured=args.captured)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / args.steps * 1e3
    lw, lt = warp.losses(), tex.losses()
    out = {"metric": f"images/sec, joint warp + texture G+D steps (both GAN pairs in one process), {H}x{W} / {S}x{S}, bs={B}/GPU",
           "value": round(B * args.steps / dt, 3), "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": f"BASELINE.json C5's shape on one GPU: warp-stage step at {H}x{W} then texture-stage step at {S}x{S} "
                                  f"(12 ROIs, L1 + VGG16 content + style), bs {B} each, train mode, fp32 storage, same kernels and "
                                  f"arithmetic as the C2 / C3 lines; one step = one optimize_parameters of each model",
                      "step_form": "hipGraph replay (swn_model_step_captured)" if args.captured else "eager launches",
                      "global_batch": B, "parallelism": "dp1"},
           "losses_finite": all(v == v and abs(v) < 1e30 for v in list(lw.values()) + list(lt.values())),
           "hbm_allocated_gb": round(ctx.bytes_allocated() / 1e9, 2)}
    print(json.dumps(out), flush=True)


def exact_f32_form(ctx, model, engine, batch, B, S, draw_labels, steps=5, warm=2):
    """Brackets the headline's arithmetic from the record alone (VERDICT r05 task 6): a SECOND model of the same shape, weights and
    batch built under SWN_SPLIT=0 -- every GEMM product an IEEE fp32 product on v_mfma_f32_32x32x2_f32, the guide's 157.3 TFLOP/s pipe --
    in the same process: its img/s and ms/step over `steps` steps, and the rel-L2 distance between the two forms' generator outputs
    (eval-mode forward of the bench batch from identical weights).  "dtype f32" of the line = fp32 storage + accumulation with
    two-fp16-plane products; this object is what the exact form costs and how far apart the two are."""
    model.forward(False, 0)
    torch.cuda.synchronize()
    ref = model.output().float().cpu()
    sd = {net: model.state_dict(net, to_cpu=True) for net in (engine.NET_G, engine.NET_D)}
    os.environ["SWN_SPLIT"] = "0"            # read per launch / per model built (conv_gemm.hip split_on, wino_pair_planes)
    m2 = None
    try:
        m2 = engine.NativeModel(ctx, "warp", B, S, S, is_train=True, dropout=0.5)
        for net, d in sd.items():
            m2.load_state_dict(net, d)
        m2.set_hyper()
        m2.set_input(0, batch["bodys"]); m2.set_input(1, batch["input_cloths"]); m2.set_input(2, batch["target_cloths"])
        m2.forward(False, 0)
        torch.cuda.synchronize()
        got = m2.output().float().cpu()
        rel = float((got - ref).norm() / ref.norm())
        worst = float((got - ref).abs().max())
        for i in range(warm):
            m2.step(draw_labels(), training=True, seed=500 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            m2.step(draw_labels(), training=True, seed=600 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        fin = all(v == v and abs(v) < 1e30 for v in m2.losses().values())
    finally:
        del os.environ["SWN_SPLIT"]
        if m2 is not None:
            m2.close()
    return {"switch": "SWN_SPLIT=0", "products": "v_mfma_f32_32x32x2_f32 (IEEE fp32 products, fp32 accumulate)", "peak_tflops": PEAK_FP32_MFMA_TFLOPS,
            "images_per_sec": round(B / dt, 2), "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warm, "losses_finite": fin,
            "fakes_rel_l2_vs_headline_form": float("%.3e" % rel), "fakes_max_abs_diff": float("%.3e" % worst),
            "note": "same process, same weights and batch as the timed steps; generator output compared in eval mode"}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it (the form the driver uses at N = 1): this process becomes the launcher --
    one rank per GPU under torch.distributed.run on 127.0.0.1 -- instead of running ONE rank and reporting n_gpus 1 under a
    --gpus N command line.  Fails loudly when fewer than N devices are visible.  The ranks' stdout / exit code pass through."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible on this box; refusing to report a "
                         f"{args.gpus}-GPU line from fewer devices")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    entry = os.environ.get("SWAPNET_BENCH_ENTRY", os.path.abspath(__file__))       # (tests: the host-simulator wrapper of this file)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), entry] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (config C2: 32)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--stage", choices=("warp", "texture", "infer", "joint"), default="warp",
                    help="warp = config C2 (the headline metric); texture = config C3 (256x256, bs 16, ROIs, "
                         "perceptual + style losses on), reported for reference")
    ap.add_argument("--precision", choices=("f32", "f16"), default="f32",
                    help="f32 (the headline, BASELINE.json C2): fp32-equivalent products from two fp16 planes per operand; "
                         "f16 (the arithmetic of C4 on however many GPUs are given): ONE fp16 plane per amax-scaled operand, "
                         "one MFMA per product, fp32 accumulation and storage -- reported with its own dtype, never as the headline")
    ap.add_argument("--grad-wire", choices=("f32", "bf16"), default=os.environ.get("SWAPNET_GRAD_WIRE", "f32"),
                    help="N > 1: the format gradient buckets travel in (parallel.GradExchange).  bf16 = BASELINE.json C4 / C5's exchange "
                         "(275 MB per generator step instead of 550 MB), rounded on the device, widened back into the fp32 arena; an "
                         "option, reported in the line, never the parity configuration")
    ap.add_argument("--captured", action="store_true",
                    help="run the step as a recorded hipGraph (swn_model_step_captured, BASELINE.json C5's captured step); "
                         "N = 1 only, bit-identical results")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--with-h2d", action="store_true",
                    help="additionally time steps that re-upload the batch from host memory every step through "
                         "the model API (set_input + step), reported as `h2d_inclusive` (never `value`)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if args.precision == "f16":        # read once by the library, before the first model is built
        os.environ["SWN_PC_PLANES"] = "1"; os.environ["SWN_WGRAD_PLANES"] = "1"
        global PC_PLANES, WGRAD_PLANES, PEAK_PC_TFLOPS
        PC_PLANES = WGRAD_PLANES = 1
        PEAK_PC_TFLOPS = PEAK_BF16_MFMA_TFLOPS
    if args.stage == "infer":
        return bench_infer(args)
    if args.stage == "joint":           # BASELINE.json C5's shape on one GPU (never the headline)
        return bench_joint(args)

    from swapnet_amd import engine, parallel, synthetic
    from swapnet_amd.modules import init_tensor

    local_rank = int(os.environ.get("SWAPNET_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (swapnet_amd has no CPU path)")
    torch.cuda.set_device(local_rank)            # before the process group: RCCL binds to the current device
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:         # never a line whose n_gpus differs from the command line's --gpus
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch one rank per GPU (or none: bench.py "
                         f"launches itself)")
    rank, world = parallel.init_from_env()
    import torch.distributed as dist

    ctx = engine.Context(device=local_rank, workspace_mb=1024)
    texture = args.stage == "texture"
    B, S = (16 if texture and args.batch == 32 else args.batch), args.size
    model = engine.NativeModel(ctx, args.stage, B, S, S, is_train=True, dropout=0.5)
    torch.manual_seed(0)                                   # identical init on every rank
    for net in (engine.NET_G, engine.NET_D) + ((engine.NET_VGG,) if texture else ()):
        sd = {}
        for name, shape in model.param_infos(net).items():
            sd[name] = torch.zeros(shape) if name.endswith(".bias") else init_tensor(torch.empty(shape), "kaiming")
        model.load_state_dict(net, sd)
    model.set_hyper(grad_scale=1.0 / world)
    if texture:
        batch = synthetic.texture_batch(B, S, S, seed=1234 + rank)
        for i, k in enumerate(("input_textures", "rois", "cloths", "target_textures")):
            model.set_input(i, batch[k])
    else:
        batch = synthetic.warp_batch(B, S, S, seed=1234 + rank)
        model.set_input(0, batch["bodys"]); model.set_input(1, batch["input_cloths"]); model.set_input(2, batch["target_cloths"])
    gG, gD = model.grad_arena(engine.NET_G), model.grad_arena(engine.NET_D)
    # SWAPNET_BENCH_RCCL1=1: 1-rank "nccl" process group and the full multi-GPU call sequence with real (identity)
    # RCCL all-reduces on the arena slices -- the N > 1 code path on a single-GPU box
    rccl1 = world == 1 and os.environ.get("SWAPNET_BENCH_RCCL1") == "1"
    if rccl1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    xchg = parallel.GradExchange(world, force=rccl1, wire=args.grad_wire)
    label_rng = torch.Generator().manual_seed(4321)        # same on every rank (SURVEY.md 8(e) caveat 2)

    def draw_labels():
        # GANLoss smooth labels: real AND fake drawn from U(0.7, 1.1) (modules/loss.py:93,102)
        return [float(torch.rand(1, generator=label_rng) * 0.4 + 0.7) for _ in range(3)]

    step_no = [0]
    # the library-owned exchange at N > 1 only with SWAPNET_NATIVE_COMM=1 (never run against a real peer: parallel.native_comm_requested),
    # by default in the 1-rank RCCL run -- agreed on by all ranks, with the
    # torch.distributed all-reduce per bucket as the other form (parallel.open_native_comm); the line says which one ran
    # (the library-owned exchange moves fp32: a bf16 wire format selects the torch form)
    native_comm = [parallel.open_native_comm(ctx) if (parallel.native_comm_requested(ctx, world) and (world > 1 or rccl1) and
                                                      args.grad_wire == "f32") else None]

    def one_step():
        lab = draw_labels()
        step_no[0] += 1
        seed = step_no[0] * 1000 + rank
        if world == 1 and not os.environ.get("SWAPNET_BENCH_PHASED") and not rccl1:
            model.step(lab, training=True, seed=seed, captured=args.captured)
            return
        if native_comm[0] is not None:
            # the library drives RCCL's all-reduce itself (swn_model_step_dp; parallel.NativeComm)
            model.step_dp(lab, training=True, seed=seed)
            return
        # (SWAPNET_BENCH_PHASED=1 runs this multi-GPU call sequence on one GPU, exchanges being no-ops, to
        # price the phased / bucketed form against the fused swn_model_step)
        model.forward(True, seed)
        model.backward_D(lab[0], lab[1])
        xchg.allreduce_mean(gD)
        model.optimizer_step(engine.NET_D)
        # generator backward in buckets (decoder + late resblocks first, encoders last): each bucket's gradients
        # travel over xGMI while the earlier layers are still being back-propagated, and each bucket's AdamW runs
        # under the next bucket's transfer
        parallel.generator_backward_with_exchange(model, lab[2], xchg)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if world > 1:          # every rank reports the device it drove (rank 0 prints them)
        mine = [rank, torch.cuda.current_device()]
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
    losses = model.losses()
    ms = dt / args.steps * 1e3
    ips = world * B * args.steps / dt
    flop_per_img = (217.8 if texture else GFLOP_PER_IMG_256) * 1e9 * (S * S) / (256 * 256)

    out = {
        "metric": ("images/sec full G+D step, texture-stage 256x256 bs=16/GPU" if texture else
                   "images/sec full G+D step, warp-stage 256x256 bs=32/GPU"),
        "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32" if args.precision == "f32" else "f16 operands (one amax-scaled plane), f32 accumulate and storage"),
        "data": "synthetic",
        "config": {"workload": (f"texture-stage G+D optimize_parameters step, {S}x{S}, bs {B}/GPU, fp32, 12 ROIs/img, "
                                f"L1 + VGG16 content + style losses, train mode, same kernels and arithmetic as the warp stage, "
                                f"TextureModule 54.5M + PatchGAN 2.8M + frozen VGG16 params, AdamW"
                                if texture else
                                f"warp-stage G+D optimize_parameters step, {S}x{S}, bs {B}/GPU, fp32 storage and "
                                f"accumulation" + ((", GEMM products on the 16-bit MFMA pipe from operand splits: forward / input-gradient "
                                                    "GEMMs as 2 fp16 planes of amax-scaled operands (22 of 24 mantissa bits, 3 of 4 terms, the dropped "
                                                    "one below 2^-22), weight-gradient GEMMs " +
                                                    ("in the same two-plane form" if WGRAD_PLANES == 2 else "as 3 bf16 planes (6 of 9 terms, dropped below 2^-24)")
                                                    if PC_PLANES == 2 else
                                                    ", REDUCED PRECISION (--precision f16): every ring-kernel GEMM multiplies ONE fp16 plane of "
                                                    "each amax-scaled operand (11 mantissa bits, one MFMA per product)" if PC_PLANES == 1 else
                                                    ", GEMM products via an exact 3 x bf16 split of both operands (6 of 9 terms, the "
                                                    "dropped ones below 2^-24) on the bf16 MFMA pipe") if SPLIT else
                                                   ", GEMM products on v_mfma_f32_32x32x2_f32") +
                                f", train mode (dropout 0.5), WarpModule 137.6M + PatchGAN 2.8M params, AdamW"),
                   "step_form": ("hipGraph replay (swn_model_step_captured)" if args.captured and world == 1 else "eager launches"),
                   "global_batch": world * B, "parallelism": f"dp{world}" + (" (1-rank RCCL exchange exercised)" if rccl1 else "")},
        "losses_finite": all(v == v and abs(v) < 1e30 for v in losses.values()),
        # proof of the launch shape for the driver's scaling table: ranks, the device each rank drives, the collective library
        "world": world,
        "ranks": ([{"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name()}] if world == 1 else None),
        "rccl_version": _rccl_version(),
        "dist_backend": (dist.get_backend() if dist.is_initialized() else None),
        "exchange": ("library-owned (swn_model_step_dp over the attached ncclAllReduce)" if native_comm[0] is not None
                     else ("torch.distributed all_reduce per bucket" if (world > 1 or rccl1) else None)),
        "exchange_wire": (args.grad_wire if (world > 1 or rccl1) else None),
        "exchange_bytes_per_step": (int(xchg.bytes_sent / max(args.steps + args.warmup, 1)) if (world > 1 or rccl1) and native_comm[0] is None else None),
        "hbm_allocated_gb": round(ctx.bytes_allocated() / 1e9, 2),      # arenas + activations (+ 2 x 1 GB split workspaces)
    }

    if world > 1:
        out["ranks"] = [{"rank": r, "device": d} for r, d in gathered]
    if rank == 0 and not args.no_roofline:
        # HIP events around every implicit-GEMM launch, on the stream they are launched on.  The timed region
        # above runs weight-gradient work on the library's second stream, where a kernel's wall time includes
        # whatever shares the GPU with it; for per-kernel durations this pass keeps everything in order on one
        # stream (same kernels, same results -- swn_ctx_set_overlap).
        ctx.set_overlap(False)
        ctx.lib.call("swn_prof_reset")
        ctx.lib.call("swn_prof_enable", 1)
        nprof = 2
        for _ in range(nprof):
            lab = draw_labels()
            model.step(lab, training=True, seed=99) if world == 1 else (
                model.forward(True, 99), model.backward_D(lab[0], lab[1]), model.backward_G(lab[2]))
        torch.cuda.synchronize()
        ctx.lib.call("swn_prof_enable", 0)
        ctx.set_overlap(True)
        import ctypes
        need = ctx.lib.dll.swn_prof_report(None, 0)
        buf = ctypes.create_string_buffer(need + 16)
        ctx.lib.dll.swn_prof_report(buf, need + 16)
        kernels = {}
        for line in buf.value.decode().splitlines():
            name, n, tms, fl = line.split()
            kernels[name] = {"launches": int(float(n)), "ms": float(tms), "flops": float(fl)}
        ctx.lib.call("swn_prof_reset")
        dom = max(kernels, key=lambda k: kernels[k]["ms"]) if kernels else None
        if dom:
            k = kernels[dom]
            ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
            gemm_ms = sum(v["ms"] for v in kernels.values()) / nprof
            traffic, traffic_src = None, None
            sp = "true" if SPLIT else "false"
            rp_name = {"conv_fwd_128x128_fast": "conv_fwd_kernel<2, 2, 2, 2, true>",
                       "conv_fwd_256x128_fast": "conv_fwd_kernel<2, 2, 4, 2, true>",
                       "conv_fwd_dma_128x128": "conv_fwd_dma_kernel<2, 2, %s>" % sp, "conv_fwd_dma_256x64": "conv_fwd_dma_kernel<4, 1, %s>" % sp,
                       "conv_fwd_dma_128x256": "conv_fwd_dma_kernel<2, 4, %s>" % sp,
                       "conv_fwd_pc_128x128": "conv_fwd_pc_kernel<4, 4, 2, 4, %d>" % PC_PLANES,
                       "conv_fwd_pc_256x64": "conv_fwd_pc_kernel<8, 2, 3, 2, %d>" % PC_PLANES,
                       "conv_fwd_pc_128x192": "conv_fwd_pc_kernel<4, 6, 2, 2, %d>" % PC_PLANES,
                       "conv_wgrad_dma_128x128": "conv_wgrad_dma_kernel<2, 2, %d>" % (0 if not SPLIT else (2 if WGRAD_PLANES == 2 else 3)),
                       "conv_wgrad_dma_256x64": "conv_wgrad_dma_kernel<4, 1, %d>" % (0 if not SPLIT else (2 if WGRAD_PLANES == 2 else 3))}.get(dom.split("[")[0], dom)
            step_hbm = None
            tnames = (("traffic_r06_texture.json", "traffic_r05_texture.json", "traffic_r04_texture.json") if texture else
                      ("traffic_r06.json", "traffic_r05.json", "traffic_r04.json"))
            for tname in tnames + ("traffic_r03b.json", "traffic_r03.json", "traffic_r02b.json", "traffic_r02.json", "traffic_r01.json"):
                tpath = os.path.join(REPO, "profiles", tname)
                if os.path.exists(tpath):
                    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
                    # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; see profiles/README.md)
                    tk = json.load(open(tpath))["kernels"]
                    # (a family = every instantiation that extends the name's template arguments: the pair-form / plane-form
                    # variants of round 4 are further arguments of the same tile's kernel; launch-weighted mean)
                    fam = [v for kname, v in tk.items() if kname == rp_name or kname.startswith(rp_name[:-1] + ",")]
                    if fam:
                        nl = sum(v["launches"] for v in fam)
                        traffic = int(sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in fam) / max(nl, 1))
                        traffic_src = "profiles/" + tname + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not collected in this run)"
                        # whole step: the pass profiles ONE step from process start, so allocation-time work (zero fills, the
                        # initial weight packing, the NCHW -> NHWC upload of the resident batch) is in the file; excluded here
                        setup = ("__amd_rocclr_fillBufferAligned", "__amd_rocclr_copyBuffer", "pack_kernel", "nchw_to_nhwc_kernel")
                        step_hbm = sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
                                       for kname, v in tk.items() if kname not in setup)
                        break
            exec_flops_step = sum(v["flops"] for v in kernels.values()) / nprof
            dense_flops_step = flop_per_img * B
            is_split = SPLIT and ("_dma_" in dom or "_pc_" in dom)
            is_pc2 = is_split and "_pc_" in dom and PC_PLANES == 2
            is_pc1 = is_split and "_pc_" in dom and PC_PLANES == 1
            is_wg = is_split and "conv_wgrad_dma" in dom
            peak = (PEAK_PC_TFLOPS if (is_pc2 or is_pc1) else
                    (round(PEAK_BF16_MFMA_TFLOPS / {1: 1.0, 2: 3.0}.get(WGRAD_PLANES, 6.0), 1) if is_wg else
                     (PEAK_SPLIT_TFLOPS if is_split else PEAK_FP32_MFMA_TFLOPS)))
            out["roofline"] = {
                "bound": "mfma", "kernel": dom, "measured": "HIP events, in-order pass (second stream off)", "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "peak_definition": ("dense fp16 MFMA peak, 2500 TFLOP/s: one MFMA per product (one fp16 plane per operand)" if is_pc1 else
                                    "fp32-equivalent FLOP/s of the 16-bit matrix pipe for this kernel's formulation: 2500 TFLOP/s dense "
                                    "fp16 / 3 fp16 MFMA products per fp32 product (two amax-scaled fp16 planes per operand, fp32 "
                                    "accumulate).  NOMINAL 2.4 GHz figure: on random operands the chip grants these loops 1.13-1.24 GHz "
                                    "(shader clock read INSIDE the kernel, tools/tile_lab.hip, profiles/tile_lab_r05*.txt) at 85-93 % "
                                    "matrix-pipe occupancy, and twelve-MFMA rounds on register-resident operands alone -- no LDS, no HBM "
                                    "-- sustain what `sustained` below reports from this very run; frac_of_sustained is against that" if is_pc2 else
                                    "fp32-equivalent FLOP/s of the bf16 matrix pipe for this kernel's formulation: 2500 TFLOP/s dense "
                                    "bf16 / 6 bf16 MFMA products per fp32 product (exact 3-way split, fp32 accumulate)" if is_split else
                                    "v_mfma_f32_32x32x2_f32 dense peak"),
                "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS, "frac_of_fp32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                # the yardstick of the round-2 / early round-3 lines (three bf16 planes, 6 MFMAs per product: 2500 / 6), so that the
                # fp32-equivalent rate can be followed across rounds although this kernel's own roofline doubled with the two-plane form
                "frac_of_bf16x6_roofline": round(ach / PEAK_SPLIT_TFLOPS, 4),
                "power_note": ("what bounds this family (DESIGN.md section 4): on operands that sit in the 256 MB Infinity Cache -- every lab figure of "
                               "round 5: back-to-back launches over one operand set -- the shipped 128x128 loop is bound by the chip's power "
                               "(tools/tile_lab.hip: 1.13-1.24 GHz at 85-93 % matrix-pipe occupancy on random operands, 2.0-2.2 GHz on zeros; 92 us "
                               "for 1024 tiles).  With the operand sets rotated so that every launch streams them from HBM, as the product's "
                               "launches do, the same loop takes 107-108 us and its fills ALONE 88 us (48 cache-resident): one 16-KB stage in "
                               "flight per workgroup, ~1.7 us per stage (profiles/tile_lab_r06_rotate.txt).  In the step the exact-fit and the "
                               "ragged launch (36 Winograd planes = 1152 tiles on 1024 places, 16 of this family's 51) both cost 0.132 us per "
                               "tile; re-issuing the ragged launch as two half launches gains nothing (profiles/ragged_split_r06.txt) and one "
                               "fp16 plane per operand -- a third of the MFMAs -- makes the family 18 % faster.  The remedies measured for "
                               "the ragged round (K-slices fused into the first round 229-433 us, five workgroups per CU 128 us, a third "
                               "ring stage does not fit four per CU: profiles/tile_lab_r06_ragged.txt) were measured cache-resident"),
                # HBM side of the same step: bytes of the PMC passes (same source as `traffic`) over this run's step time, against
                # the 6.3 TB/s the guide measures as achievable (8 TB/s spec)
                "step_hbm_bytes": step_hbm,
                "step_hbm_frac": (round(step_hbm / (ms * 1e-3) / 6.3e12, 4) if step_hbm else None),
                "avg_launch_ms": round(k["ms"] / k["launches"], 4), "launches_per_step": k["launches"] // nprof,
                "step_frac_executed": round(exec_flops_step / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                "executed_tflop_per_step": round(exec_flops_step / 1e12, 3),
                "algorithmic_speedup": round(dense_flops_step / exec_flops_step, 3),
                "dense_equivalent_frac": round(dense_flops_step / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                "gemm_ms_per_step": round(gemm_ms, 2),
                # the yardstick that does not move with the formulation: 16-bit MFMA FLOPs actually EXECUTED
                # (3 per fp32 product in the two-plane kernels, 6 in the three-plane ones) over the nominal 2.5 PFLOP/s dense
                # peak -- for the dominant kernel's own launches, and for the whole step over the step time
                "pipe_util_nominal": round(ach * mfma16_per_product(dom) / PEAK_BF16_MFMA_TFLOPS, 4),
                "pipe_util_nominal_step": round(sum(v["flops"] * mfma16_per_product(n) for n, v in kernels.items()) / nprof
                                                / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                "all_gemm_kernels": {n: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                         "ms_per_step": round(v["ms"] / nprof, 3)} for n, v in sorted(kernels.items())},
            }
    if rank == 0 and "roofline" in out:
        # what the matrix pipe of THIS chip sustains for the ring kernels' instruction mix with the operands in registers
        # (swn_probe_mfma: 1024 workgroups x 4 waves x rounds of 12 v_mfma_f32_32x32x16_f16, clock read inside the kernel)
        import ctypes
        sus = {}
        for tag, zeros in (("random_operands", 0), ("zero_operands", 1)):
            o4 = (ctypes.c_float * 4)()
            ctx.lib.call("swn_probe_mfma", ctx.handle, zeros, 4096, o4)
            sus[tag] = {"fp16_tflops": round(o4[0], 1), "shader_clock_ghz": round(o4[1], 3), "ms": round(o4[2], 3),
                        "pipe_occupancy_at_that_clock": round(o4[3], 3)}
        r = out["roofline"]
        r["sustained"] = sus
        per_product = mfma16_per_product(r["kernel"])
        if sus["random_operands"]["fp16_tflops"] > 0 and per_product:
            r["sustained"]["fp32_equivalent_tflops"] = round(sus["random_operands"]["fp16_tflops"] / per_product, 1)
            r["frac_of_sustained"] = round(r["achieved"] * per_product / sus["random_operands"]["fp16_tflops"], 4)
    if rank == 0 and (world > 1 or os.environ.get("SWAPNET_BENCH_PHASED") or rccl1):
        # measured back-propagation time and gradient bytes of each exchange bucket (what engine.cpp's bucket boundaries are
        # sized on): HIP events around swn_model_backward_G_part, no exchange in between
        lab = draw_labels()
        model.forward(True, 7)
        model.backward_D(lab[0], lab[1])
        torch.cuda.synchronize()
        evs, sizes = [torch.cuda.Event(enable_timing=True)], []
        evs[0].record()
        for part in range(model.backward_G_parts()):
            off, cnt = model.backward_G_part(lab[2], part)
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); sizes.append(cnt * 4)
        torch.cuda.synchronize()
        out["dp_buckets"] = [{"grad_bytes": sizes[i], "backward_ms": round(evs[i].elapsed_time(evs[i + 1]), 3)} for i in range(len(sizes))]
    if rank == 0 and world == 1 and args.with_h2d and not texture:
        # the boundary hands over host buffers (warp_model.py:99-104): PCIe-inclusive rate, one-hot fp32
        # cloths (the reference's format) and integer label maps (device-side one-hot)
        host = [batch["bodys"], batch["input_cloths"], batch["target_cloths"]]
        labels = [t.argmax(1).to(torch.int32) * (t.sum(1) > 0) for t in host[1:]]
        pinned = [t.pin_memory() for t in host]          # DataLoader(pin_memory=True)
        res = {}
        for name, feed in (("onehot_fp32", lambda: [model.set_input(i, t) for i, t in enumerate(host)]),
                           ("onehot_fp32_pinned", lambda: [model.set_input(i, t) for i, t in enumerate(pinned)]),
                           ("label_maps", lambda: (model.set_input(0, host[0]), model.set_input_labels(1, labels[0]),
                                                   model.set_input_labels(2, labels[1])))):
            for _ in range(2):
                feed(); one_step()
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                feed(); one_step()
            fence()
            d1 = (time.perf_counter() - t1) / args.steps
            res[name] = {"images_per_sec": round(B / d1, 2), "ms_per_step": round(d1 * 1e3, 3)}
        # what the box's host->device link itself delivers (plain torch copies of one 256 MB buffer, best of 5): the
        # PCIe-inclusive rates above are bounded by this, not by the library
        raw = {}
        probe = torch.rand(64 << 20, dtype=torch.float32)          # touched pages with real data (an untouched
        dst = torch.empty_like(probe, device="cuda")               # torch.empty maps the zero page and copies "fast")
        for name, src in (("pageable", probe), ("pinned", probe.pin_memory())):
            best = 0.0
            for _ in range(5):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
                best = max(best, probe.numel() * 4 / (time.perf_counter() - t1) / 1e9)
            raw[name] = round(best, 2)
        res["raw_h2d_GBps"] = raw
        res["h2d_bytes_per_step"] = {"onehot_fp32": sum(t.numel() * 4 for t in host),
                                     "label_maps": host[0].numel() * 4 + sum(t.numel() * 4 for t in labels)}
        out["h2d_inclusive"] = res
    if rank == 0 and not args.no_roofline:      # (the PMC passes of profiles/ run with --no-roofline: exactly the timed steps)
        # which kernel every layer launched (swn_route_trace): the digest the parity tests compare their own runs with
        # (tests/backends.py assert_default_routing), and the switches that were set in this process
        import hashlib
        ctx.route_trace(True)
        lab = draw_labels()
        model.forward(True, 3); model.backward_D(lab[0], lab[1]); model.optimizer_step(engine.NET_D)
        model.backward_G(lab[2]); model.optimizer_step(engine.NET_G)
        torch.cuda.synchronize()
        ctx.route_trace(False)
        route = ctx.route_report()
        out["config"]["routing"] = {"launch_lines": len(route), "sha16": hashlib.sha256("\n".join(route).encode()).hexdigest()[:16],
                                    "switches": sorted(k + "=" + v for k, v in os.environ.items() if k.startswith("SWN_"))}
    if rank == 0 and world == 1 and "roofline" in out and not texture and SPLIT and args.precision == "f32" and not args.captured:
        out["roofline"]["exact_f32_form"] = exact_f32_form(ctx, model, engine, batch, B, S, draw_labels)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_texture() if texture else cpu_baseline()
    # the JSON line is the LAST thing on stdout: RCCL writes its version banner through C stdio, which (not a tty) sits in a
    # buffer until the process exits -- flush it, and tear the process group down, before printing
    import ctypes
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    if world > 1 or rccl1:
        dist.barrier()
        if native_comm[0] is not None:        # the library's communicator goes before the process group it was bootstrapped over
            native_comm[0].close()
            native_comm[0] = None
        dist.destroy_process_group()
        libc.fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
