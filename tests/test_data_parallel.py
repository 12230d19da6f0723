"""Data-parallel path on CPU: 2 gloo ranks x 1 sample each, gradient arenas
all-reduced between backward and optimizer step, must equal one process with both samples
(every layer is per-sample and every loss a batch mean, so DP is exact up to fp32 summation
order).  Runs the real engine + C-ABI through the CI host simulator."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.small_channel_winograd      # tests/conftest.py: small shapes on the Winograd forms

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine, parallel
    from tests import backends
    r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    ctx = backends.hostsim_ctx()
    torch.manual_seed(0)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    full = O.synth_warp_batch(world, 64, 64, seed=1234)
    m = engine.NativeModel(ctx, "warp", 1, 64, 64, is_train=True)
    m.load_state_dict(0, G); m.load_state_dict(1, D)
    m.set_hyper(grad_scale=1.0 / world)
    for i, t in enumerate(full):
        m.set_input(i, t[rank:rank + 1])
    x = parallel.GradExchange(world)
    lab = [0.9, 0.8, 1.0]
    m.forward(False, 0)
    m.backward_D(lab[0], lab[1])
    x.allreduce_mean(m.grad_arena(engine.NET_D))
    m.optimizer_step(engine.NET_D)
    # bucketed backward: exchange of bucket k under the back-propagation of bucket k+1, AdamW of a bucket as soon as its
    # all-reduce landed (swn_model_optimizer_step_range) -- the production schedule of bench.py / base_gan
    assert m.backward_G_parts() >= 2
    parallel.generator_backward_with_exchange(m, lab[2], x)
    assert m.optim_step_count(engine.NET_G) == 1
    if rank == 0:
        torch.save({"G": m.state_dict(0, to_cpu=True), "D": m.state_dict(1, to_cpu=True)}, os.path.join(out_dir, "dp.pt"))
    # replicas stay identical
    wsum = m.weight_arena(0).clone()
    dist.all_reduce(wsum)
    assert torch.allclose(wsum / world, m.weight_arena(0), rtol=0, atol=0)
    dist.barrier()
    dist.destroy_process_group()


def _wire_worker(rank, world, port, out_dir, wire):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine, parallel
    from tests import backends
    parallel.init_from_env(backend="gloo")
    ctx = backends.hostsim_ctx()
    torch.manual_seed(0)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    full = O.synth_warp_batch(world, 64, 64, seed=1234)
    m = engine.NativeModel(ctx, "warp", 1, 64, 64, is_train=True)
    m.load_state_dict(0, G); m.load_state_dict(1, D)
    m.set_hyper(grad_scale=1.0 / world)
    for i, t in enumerate(full):
        m.set_input(i, t[rank:rank + 1])
    x = parallel.GradExchange(world, wire=wire)
    lab = [0.9, 0.8, 1.0]
    m.forward(False, 0)
    m.backward_D(lab[0], lab[1])
    x.allreduce_mean(m.grad_arena(engine.NET_D))
    gD = m.grad_arena(engine.NET_D).clone()
    m.optimizer_step(engine.NET_D)
    parallel.generator_backward_with_exchange(m, lab[2], x)
    gG = m.grad_arena(engine.NET_G).clone()
    # replicas hold the same reduced gradients and the same weights, whatever the wire format
    for t in (gG, m.weight_arena(0)):
        s2 = t.clone(); dist.all_reduce(s2)
        assert torch.equal(s2 / world, t)
    if rank == 0:
        torch.save({"gG": gG, "gD": gD, "G": m.state_dict(0, to_cpu=True), "sent": x.bytes_sent}, os.path.join(out_dir, "wire_%s.pt" % wire))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_wire_format_against_the_fp32_exchange(tmp_path):
    """BASELINE.json C4 / C5 name a bf16 gradient exchange (275 MB instead of 550 MB per generator step).
    GradExchange(wire="bf16") / SWAPNET_GRAD_WIRE=bf16: buckets travel as bfloat16 and land widened in the fp32 arena; AdamW and the
    weights stay fp32.  Two gloo ranks, both formats from the same state: half the bytes, the reduced gradients within bfloat16's
    rounding (2^-8) of the fp32 exchange's -- and not identical to it (the format really was on the wire) -- the post-step weights
    within 2e-3 (Adam's first step is lr * sign(g): only elements whose sign the rounding flips move, by 2 lr), replicas bit-identical
    to each other.  An option: the parity configuration exchanges fp32."""
    from tests import backends
    backends.build_hostsim()
    out = {}
    for i, wire in enumerate(("f32", "bf16")):
        port = 36800 + os.getpid() % 2000 + 3 * i
        mp.spawn(_wire_worker, args=(2, port, str(tmp_path), wire), nprocs=2, join=True)
        out[wire] = torch.load(os.path.join(tmp_path, "wire_%s.pt" % wire))
    assert out["bf16"]["sent"] * 2 == out["f32"]["sent"]
    for k in ("gG", "gD"):
        a, b = out["bf16"][k].double(), out["f32"][k].double()
        e = float((a - b).norm() / b.norm())
        assert 1e-5 < e < 6e-3, (k, e)
        assert torch.equal(out["bf16"][k], out["bf16"][k].to(torch.bfloat16).float())       # every element IS a bfloat16 value
    for k, v in out["f32"]["G"].items():
        if k.endswith(".bias") and "resblocks" in k:
            continue
        e = float((out["bf16"]["G"][k] - v).norm() / (v.norm() + 1e-30))
        assert e < 2e-3, (k, e)


def test_two_rank_gloo_step_equals_single_process_big_batch(tmp_path):
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine
    from tests import backends
    backends.build_hostsim()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    dp = torch.load(os.path.join(tmp_path, "dp.pt"))
    ctx = backends.hostsim_ctx()
    torch.manual_seed(0)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    full = O.synth_warp_batch(2, 64, 64, seed=1234)
    m = engine.NativeModel(ctx, "warp", 2, 64, 64, is_train=True)
    m.load_state_dict(0, G); m.load_state_dict(1, D); m.set_hyper()
    for i, t in enumerate(full):
        m.set_input(i, t)
    m.step([0.9, 0.8, 1.0], training=False, seed=0)
    for net, key in ((0, "G"), (1, "D")):
        sd = m.state_dict(net, to_cpu=True)
        for k, v in sd.items():
            if k.endswith(".bias") and ("resblocks" in k or k.startswith(("model.2.", "model.5.", "model.8."))):
                continue                       # round-off-only gradients
            err = float((dp[key][k] - v).norm() / (v.norm() + 1e-30))
            assert err < 2e-4, (key, k, err)


def _texture_worker(rank, world, port, out_dir, lambda_style):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine, parallel
    from tests import backends
    from tests.test_texture_step import vgg_state_dict
    parallel.init_from_env(backend="gloo")
    ctx = backends.hostsim_ctx()
    torch.manual_seed(0)
    G, D, vgg = O.texture_module_params(img_size=64), O.patchgan_params(22), O.vgg16_feature_params()
    full = O.synth_texture_batch(world, 64, 64, seed=77)
    m = engine.NativeModel(ctx, "texture", 1, 64, 64, is_train=True)
    m.load_state_dict(0, G); m.load_state_dict(1, D); m.load_state_dict(2, vgg_state_dict(m, vgg))
    m.set_hyper(grad_scale=1.0 / world, lambda_style=lambda_style)
    for i, t in enumerate(full):
        m.set_input(i, t[rank:rank + 1])
    x = parallel.GradExchange(world)
    lab = [0.9, 0.8, 1.0]
    m.forward(False, 0)
    m.backward_D(lab[0], lab[1])
    x.allreduce_mean(m.grad_arena(engine.NET_D))
    m.optimizer_step(engine.NET_D)
    if lambda_style != 0:
        parallel.gather_style_context(m, full[3][rank:rank + 1])
    parallel.generator_backward_with_exchange(m, lab[2], x)
    if rank == 0:
        torch.save({"G": m.state_dict(0, to_cpu=True), "gG": m.state_dict(0, which=engine.W_GRAD, to_cpu=True)},
                   os.path.join(out_dir, "dp_tex_%g.pt" % lambda_style))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lambda_style", [0.0, 1e-8])
def test_two_rank_texture_step_vs_big_batch(tmp_path, lambda_style):
    """the texture stage under data parallelism equals the one-process big-batch step (to summation order)
    with the style term off AND on: the style term's Gram spans the whole batch, so the ranks all-gather their 3-channel
    images and the library evaluates it globally (parallel.gather_style_context).  (Without that, a per-rank Gram moved
    generator gradient tensors by 12 % in this very test.)"""
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine
    from tests import backends
    from tests.test_texture_step import noise_bias, vgg_state_dict
    backends.build_hostsim()
    port = 31500 + os.getpid() % 2000 + int(lambda_style > 0)
    mp.spawn(_texture_worker, args=(2, port, str(tmp_path), lambda_style), nprocs=2, join=True)
    dp = torch.load(os.path.join(tmp_path, "dp_tex_%g.pt" % lambda_style))
    ctx = backends.hostsim_ctx()
    torch.manual_seed(0)
    G, D, vgg = O.texture_module_params(img_size=64), O.patchgan_params(22), O.vgg16_feature_params()
    full = O.synth_texture_batch(2, 64, 64, seed=77)
    m = engine.NativeModel(ctx, "texture", 2, 64, 64, is_train=True)
    m.load_state_dict(0, G); m.load_state_dict(1, D); m.load_state_dict(2, vgg_state_dict(m, vgg))
    m.set_hyper(lambda_style=lambda_style)
    for i, t in enumerate(full):
        m.set_input(i, t)
    m.forward(False, 0); m.backward_D(0.9, 0.8); m.optimizer_step(1); m.backward_G(1.0)
    gG = m.state_dict(0, which=engine.W_GRAD, to_cpu=True)
    keys = list(G.keys())
    worst = max(float((dp["gG"][k] - v).norm() / (v.norm() + 1e-30)) for k, v in gG.items() if not noise_bias(k, keys))
    assert worst < 2e-4, worst
    m.close()


@pytest.mark.parametrize("backend", [pytest.param("sim", id="hostsim"),
                                     pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("stage", ["warp", "texture"])
def test_bucketed_generator_backward_equals_monolithic(stage, backend):
    """swn_model_backward_G_part over all buckets == swn_model_backward_G, bit for bit, for both
    generators; each bucket's reported arena range is already final when its part returns (checked by
    snapshotting the range right after the part and comparing with the finished gradient)."""
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine
    from tests import backends
    ctx = backends.gpu_ctx() if backend == "gpu" else backends.hostsim_ctx()
    torch.manual_seed(0)
    D = O.patchgan_params(22)
    if stage == "warp":
        G = O.warp_module_params()
        batch = O.synth_warp_batch(1, 64, 64, seed=3)
    else:
        G = O.texture_module_params(img_size=64)
        batch = O.synth_texture_batch(1, 64, 64, seed=3)
    m = engine.NativeModel(ctx, stage, 1, 64, 64, is_train=True)
    try:
        m.load_state_dict(0, G); m.load_state_dict(1, D); m.set_hyper()
        if stage == "texture":
            vgg = O.vgg16_feature_params()
            names = list(m.param_infos(engine.NET_VGG).keys())
            m.load_state_dict(engine.NET_VGG, {names[2 * i + j]: t for i, wb in enumerate(vgg) for j, t in enumerate(wb)})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        m.forward(False, 0)
        m.backward_G(0.9)
        ref = m.grad_arena(engine.NET_G).clone()
        m.grad_arena(engine.NET_G).zero_()
        m.forward(False, 0)
        end, snaps = ref.numel(), []
        for part in range(m.backward_G_parts()):
            off, cnt = m.backward_G_part(0.9, part)
            assert cnt > 0 and off + cnt == end
            snaps.append((off, cnt, m.grad_arena(engine.NET_G)[off:off + cnt].clone()))
            end = off
        assert end == 0
        got = m.grad_arena(engine.NET_G)
        assert torch.equal(got.cpu(), ref.cpu())
        for off, cnt, snap in snaps:
            assert torch.equal(snap.cpu(), ref[off:off + cnt].cpu()), (off, cnt)
    finally:
        m.close()


@pytest.mark.gpu
def test_second_stream_leaves_results_bit_identical():
    """swn_ctx_set_overlap: the weight-gradient side stream / operand prefetch reorder work in time only.
    Two steps from the same state, overlap on vs off, must give bitwise equal weights and losses."""
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine
    from tests import backends
    ctx = backends.gpu_ctx()
    torch.manual_seed(1)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(4, 128, 128, seed=11)
    m = engine.NativeModel(ctx, "warp", 4, 128, 128, is_train=True)
    out = []
    try:
        for on in (True, False, True):
            ctx.set_overlap(on)
            backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
            for i, t in enumerate(batch):
                m.set_input(i, t)
            for s in range(2):
                m.step([0.9, 0.8, 1.0], training=True, seed=5 + s)
            out.append((m.losses(), m.weight_arena(0).clone().cpu(), m.weight_arena(1).clone().cpu()))
    finally:
        ctx.set_overlap(True)
        m.close()
    for o in out[1:]:
        assert o[0] == out[0][0]
        assert torch.equal(o[1], out[0][1]) and torch.equal(o[2], out[0][2])


@pytest.mark.gpu
def test_one_rank_rccl_exchange_equals_the_fused_step():
    """The N > 1 call sequence with REAL RCCL collectives (backend "nccl", world size 1: every all-reduce is an identity)
    on the zero-copy arena slices: pointer wrapping, the ordering between torch's collective stream, the library's compute
    stream and its weight-gradient side stream, and the ranged AdamW steps must reproduce swn_model_step bit for bit."""
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine, parallel
    from tests import backends
    ctx = backends.gpu_ctx()
    if not dist.is_initialized():
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200))
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(1)
        G, D = O.warp_module_params(), O.patchgan_params(22)
        batch = O.synth_warp_batch(4, 128, 128, seed=11)
        m = engine.NativeModel(ctx, "warp", 4, 128, 128, is_train=True)
        out = []
        for mode in ("fused", "rccl"):
            backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
            for i, t in enumerate(batch):
                m.set_input(i, t)
            for step in range(2):
                lab = [0.9, 0.8, 1.0]
                if mode == "fused":
                    m.step(lab, training=True, seed=5 + step)
                else:
                    x = parallel.GradExchange(1, force=True)
                    m.forward(True, 5 + step)
                    m.backward_D(lab[0], lab[1])
                    x.allreduce_mean(m.grad_arena(engine.NET_D))
                    m.optimizer_step(engine.NET_D)
                    parallel.generator_backward_with_exchange(m, lab[2], x)
            out.append((m.losses(), m.weight_arena(0).clone().cpu(), m.weight_arena(1).clone().cpu(), m.optim_step_count(0)))
        m.close()
        assert out[0][0] == out[1][0] and out[0][3] == out[1][3] == 2
        assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    finally:
        dist.destroy_process_group()


# ---- library-owned exchange: swn_ctx_attach_comm + swn_model_step_dp (round 4) --------------------------------------------------
def test_library_owned_exchange_with_one_rank_equals_the_fused_step():
    """swn_model_step_dp with a one-rank communicator attached (callback form of parallel.NativeComm: every all-reduce is an
    identity) must reproduce swn_model_step bit for bit -- weights, both Adam moments, step counters -- over two steps, and the
    collectives it issues are one over D's arena plus one per generator bucket, whose ranges tile the generator's arena."""
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine, parallel
    from tests import backends
    ctx = backends.hostsim_ctx(fresh=True)
    torch.manual_seed(1)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(1, 64, 64, seed=11)
    m = engine.NativeModel(ctx, "warp", 1, 64, 64, is_train=True)
    comm = None
    try:
        out = []
        for mode in ("fused", "native"):
            backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
            for i, t in enumerate(batch):
                m.set_input(i, t)
            if mode == "native":
                with pytest.raises(Exception):
                    m.step_dp([0.9, 0.8, 1.0])                     # nothing attached yet: refused, not silently local
                comm = parallel.NativeComm(ctx, backend="gloo")
            for step in range(2):
                (m.step if mode == "fused" else m.step_dp)([0.9, 0.8, 1.0], training=True, seed=5 + step)
            out.append((m.losses(), m.weight_arena(0).clone(), m.weight_arena(1).clone(),
                        m.state_dict(0, which=engine.W_EXP_AVG_SQ, to_cpu=True), m.optim_step_count(0), m.optim_step_count(1)))
        assert out[0][0] == out[1][0]
        assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
        assert all(torch.equal(out[0][3][k], out[1][3][k]) for k in out[0][3])
        assert out[0][4:] == out[1][4:] == (2, 2)
        nG, nD, parts = m.grad_arena(0).numel(), m.grad_arena(1).numel(), m.backward_G_parts()
        per_step = comm.calls[:1 + parts]
        assert len(comm.calls) == 2 * (1 + parts) and per_step[0] == nD and sum(per_step[1:]) == nG, (comm.calls, nG, nD)
    finally:
        if comm:
            comm.close()
        m.close()
    with pytest.raises(Exception):
        m2 = engine.NativeModel(ctx, "warp", 1, 64, 64, is_train=True)
        try:
            m2.step_dp([0.9, 0.8, 1.0])                            # detached again
        finally:
            m2.close()


def _native_worker(rank, world, port, out_dir, stage, lambda_style):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine, parallel
    from tests import backends
    parallel.init_from_env(backend="gloo")
    ctx = backends.hostsim_ctx()
    torch.manual_seed(0)
    D = O.patchgan_params(22)
    if stage == "warp":
        G, full = O.warp_module_params(), O.synth_warp_batch(world, 64, 64, seed=1234)
    else:
        from tests.test_texture_step import vgg_state_dict
        G, full = O.texture_module_params(img_size=64), O.synth_texture_batch(world, 64, 64, seed=77)
        vgg = O.vgg16_feature_params()
    lab = [0.9, 0.8, 1.0]
    res = []
    for mode in ("torch", "native"):
        m = engine.NativeModel(ctx, stage, 1, 64, 64, is_train=True)
        m.load_state_dict(0, G); m.load_state_dict(1, D)
        if stage == "texture":
            m.load_state_dict(2, vgg_state_dict(m, vgg))
            m.set_hyper(grad_scale=1.0 / world, lambda_style=lambda_style)
        else:
            m.set_hyper(grad_scale=1.0 / world)
        for i, t in enumerate(full):
            m.set_input(i, t[rank:rank + 1])
        comm = None
        for step in range(2):
            if mode == "torch":                  # the phased calls around torch.distributed collectives (GradExchange)
                x = parallel.GradExchange(world)
                m.forward(False, 0)
                m.backward_D(lab[0], lab[1])
                x.allreduce_mean(m.grad_arena(engine.NET_D))
                m.optimizer_step(engine.NET_D)
                if stage == "texture" and lambda_style != 0:
                    parallel.gather_style_context(m, full[3][rank:rank + 1])
                parallel.generator_backward_with_exchange(m, lab[2], x)
            else:                                # one call, the library drives the attached all-reduce itself
                comm = comm or parallel.NativeComm(ctx, backend="gloo")
                if stage == "texture" and lambda_style != 0:
                    m.forward(False, 0)
                    parallel.gather_style_context(m, full[3][rank:rank + 1])
                    m.step_dp(lab, training=False, seed=0, after_forward=True)
                else:
                    m.step_dp(lab, training=False, seed=0)
        res.append((m.weight_arena(0).clone(), m.weight_arena(1).clone(), m.losses(), m.optim_step_count(0)))
        if comm:
            comm.close()
        m.close()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), "native exchange differs from the torch path"
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] == 2
    wsum = res[1][0].clone()
    dist.all_reduce(wsum)
    assert torch.equal(wsum / world, res[1][0])                       # replicas stay identical
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stage,lambda_style", [("warp", 0.0), ("texture", 1e-8)])
def test_two_rank_library_owned_exchange_equals_the_torch_path(tmp_path, stage, lambda_style):
    """2 gloo ranks x 1 sample, two steps: swn_model_step_dp (exchange driven by the library through the attached all-reduce,
    AdamW of a bucket behind its all-reduce) gives bitwise the weights of the phased torch.distributed path -- which
    test_two_rank_gloo_step_equals_single_process_big_batch holds against the one-process big-batch step.  Texture stage with the
    style term: forward by the caller, global-batch style context, then swn_model_step_dp(after_forward)."""
    from tests import backends
    backends.build_hostsim()
    port = 33500 + os.getpid() % 2000 + (7 if stage == "texture" else 0)
    mp.spawn(_native_worker, args=(2, port, str(tmp_path), stage, lambda_style), nprocs=2, join=True)


def _agree_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from swapnet_amd import parallel
    from tests import backends
    parallel.init_from_env(backend="gloo")
    ctx = backends.hostsim_ctx()
    # (a) every rank can bring its communicator up: all get one
    comm = parallel.open_native_comm(ctx)
    assert comm is not None and ctx.native_comm is comm
    comm.close()
    assert ctx.native_comm is None
    # (b) ONE rank cannot (here: rank 1's constructor raises): every rank must end up without one, the able rank's closed again
    real = parallel.NativeComm
    made = []

    class Flaky(real):
        def __init__(self, c, backend=None):
            if rank == 1:
                raise RuntimeError("ncclCommInitRank failed: 5 (simulated)")
            super().__init__(c, backend)
            made.append(self)
    parallel.NativeComm = Flaky
    try:
        got = parallel.open_native_comm(ctx)
    finally:
        parallel.NativeComm = real
    assert got is None and ctx.native_comm is None
    if rank == 0:
        assert len(made) == 1 and made[0].comm is None              # brought up, then closed by the agreement
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_exchange_form(tmp_path):
    """parallel.open_native_comm (round 5: the library-owned exchange is the default on a GPU box): if ANY rank cannot bring its
    communicator up, EVERY rank falls back to the torch.distributed all-reduce per bucket -- a rank stepping through
    swn_model_step_dp while another waits in dist.all_reduce would hang the job.  Two gloo ranks, the callback form of NativeComm
    on the host simulator; on the MI355X the same agreement ran with two ranks on one device, where RCCL refuses the second
    (profiles/two_ranks_one_gpu_r05.txt)."""
    from tests import backends
    backends.build_hostsim()
    port = 34600 + os.getpid() % 2000
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _asym_worker(rank, world, port, out_dir, inject):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), SWAPNET_TEST_RCCL_FAIL=inject)
    torch.set_num_threads(2)
    from swapnet_amd import parallel
    from tests import backends
    parallel.init_from_env(backend="gloo")
    ctx = backends.hostsim_ctx()
    real = parallel.NativeComm

    class RcclBringUp(real):                  # the REAL _init_rccl (its collectives included) on a context that is not a device build
        def __init__(self, c, backend=None):
            super().__init__(c, backend="rccl")
    parallel.NativeComm = RcclBringUp
    try:
        got = parallel.open_native_comm(ctx)
    finally:
        parallel.NativeComm = real
    assert got is None and getattr(ctx, "native_comm", None) is None
    dist.barrier()                            # the process group is still in step: no rank is stuck in a collective the other skipped
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("inject", ["rank0-before-broadcast", "rank1-before-init"])
def test_asymmetric_bring_up_failure_does_not_hang_the_ranks(tmp_path, inject):
    """NativeComm._init_rccl has collectives of its own (the uid broadcast, ncclCommInitRank).  If rank 0 fails before the
    broadcast, or a non-zero rank before ncclCommInitRank, the others used to wait there forever.  The bring-up now agrees stage by
    stage (NativeComm._agree): every rank enters or skips each collective together, and all end on the torch form.  Two gloo ranks
    drive the real _init_rccl with the failure injected (SWAPNET_TEST_RCCL_FAIL); a hang fails by timeout."""
    from tests import backends
    backends.build_hostsim()
    port = 35700 + os.getpid() % 2000 + (11 if inject.startswith("rank1") else 0)
    ctxm = mp.spawn(_asym_worker, args=(2, port, str(tmp_path), inject), nprocs=2, join=False)
    import time as _t
    t0 = _t.time()
    while not ctxm.join(timeout=5):
        if _t.time() - t0 > 240:
            for p in ctxm.processes:
                p.kill()
            raise AssertionError("ranks hung in the bring-up (" + inject + ")")
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


@pytest.mark.gpu
def test_one_rank_native_rccl_exchange_equals_the_fused_step():
    """parallel.NativeComm(backend="rccl") at world size 1: ncclCommInitRank through ctypes on the RCCL the process holds, RCCL's own
    ncclAllReduce driven by the library on its exchange stream (swn_model_step_dp), AdamW per bucket on that stream -- bit-identical
    to swn_model_step over two training-mode steps.  Written without a GPU (round 4): opt-in until it has run once."""
    from oracle import swapnet_oracle as O
    from swapnet_amd import engine, parallel
    from tests import backends
    ctx = backends.gpu_ctx()
    torch.manual_seed(1)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(4, 128, 128, seed=11)
    m = engine.NativeModel(ctx, "warp", 4, 128, 128, is_train=True)
    comm = None
    try:
        out = []
        for mode in ("fused", "native"):
            backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
            for i, t in enumerate(batch):
                m.set_input(i, t)
            if mode == "native":
                comm = parallel.NativeComm(ctx, backend="rccl")
            for step in range(2):
                (m.step if mode == "fused" else m.step_dp)([0.9, 0.8, 1.0], training=True, seed=5 + step)
            out.append((m.losses(), m.weight_arena(0).clone().cpu(), m.weight_arena(1).clone().cpu(), m.optim_step_count(0)))
        assert out[0][0] == out[1][0] and out[0][3] == out[1][3] == 2
        assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    finally:
        if comm:
            comm.close()
        m.close()


@pytest.mark.parametrize("native", [False, True], ids=["torch-exchange", "library-owned-exchange"])
def test_bench_py_launch_reduce_and_report_path_for_two_ranks(native):
    """`bench.py --gpus N` for N > 1 exactly as the driver launches it (python -m torch.distributed.run --nproc-per-node N ...): rank /
    device selection from the environment, process group, per-rank batches, the bucketed exchange inside the timed loop, the barrier +
    MAX-over-ranks timing, the gathered rank table and the ONE JSON line of rank 0.  No 1-GPU box executes these lines, so they run
    here with two gloo ranks on the host simulator (tests/bench_on_hostsim.py runs the UNCHANGED bench.py with the simulator standing
    in for the GPU) -- what is checked is the launch contract and the report, not a rate."""
    import json
    import subprocess
    import sys
    from tests import backends
    backends.build_hostsim()
    env = {k: v for k, v in os.environ.items() if not k.startswith("SWN_")}
    env.update(SWAPNET_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    env["SWAPNET_NATIVE_COMM"] = "1" if native else "0"          # (the default on a GPU box is the library-owned exchange: parallel.native_comm_requested)
    port = 29900 + os.getpid() % 90 + (5 if native else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(backends.REPO, "tests", "bench_on_hostsim.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "1", "--size", "64", "--no-roofline", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=backends.REPO)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["unit"] == "images/sec" and d["losses_finite"] is True
    assert sorted(r["rank"] for r in d["ranks"]) == [0, 1] and sorted(r["device"] for r in d["ranks"]) == [0, 1]
    assert d["config"]["global_batch"] == 2 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 1 * 2 / (d["ms_per_step"] * 2e-3)) <= 0.02 * d["value"]       # whole-job rate: world x B x steps / time
    assert d["dist_backend"] == "gloo"
    assert d["exchange"].startswith("library-owned" if native else "torch.distributed")


def test_bench_py_launches_its_own_ranks_when_no_launcher_is_around_it():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the form the driver uses at N = 1; VERDICT r05 missing #4): bench.py
    becomes the launcher (torch.distributed.run, one rank per device, 127.0.0.1), the line says n_gpus 2 -- and with fewer devices visible
    than --gpus it refuses loudly instead of reporting an N-GPU line from one rank.  Two gloo ranks on the host simulator."""
    import json
    import subprocess
    import sys
    from tests import backends
    backends.build_hostsim()
    env = {k: v for k, v in os.environ.items() if not k.startswith("SWN_") and k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SWAPNET_DIST_BACKEND="gloo", OMP_NUM_THREADS="4", SWAPNET_NATIVE_COMM="0")
    cmd = [sys.executable, os.path.join(backends.REPO, "tests", "bench_on_hostsim.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--batch", "1", "--size", "64", "--no-roofline", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=backends.REPO)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world"] == 2 and sorted(r["device"] for r in d["ranks"]) == [0, 1] and d["config"]["parallelism"] == "dp2"
    few = subprocess.run(cmd, env=dict(env, SWAPNET_SIM_DEVICES="1"), capture_output=True, text=True, timeout=300, cwd=backends.REPO)
    assert few.returncode != 0 and "only 1 HIP device(s) visible" in few.stderr and not [l for l in few.stdout.splitlines() if l.startswith("{")]
    # a launcher whose world size disagrees with --gpus is refused too (never n_gpus != --gpus)
    bad = subprocess.run(cmd[:2] + ["--gpus", "1"] + cmd[4:], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_PORT="29977"), capture_output=True,
                         text=True, timeout=300, cwd=backends.REPO)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr
