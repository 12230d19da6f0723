"""Parity in the configuration bench.py times: TRAINING mode (dropout 0.5 at the
reference's 7 warp / 3 texture sites, modules/layers.py:22-23,136 and pix2pix_modules.py:251-252), the full
batch sizes of BASELINE.json C2 / C3, and gradient tolerances measured against an fp64 evaluation.

How a training-mode step can be compared value for value: the library exports, per dropout site, the keep/scale
factor its forward AND backward kernels apply for a given seed (swn_model_dropout_mask).  The oracle replays those
tensors at its nn.Dropout call sites (oracle.MaskReplay) -- autograd then multiplies the gradient by the same
tensor, which is nn.Dropout's definition -- so a wrong scale, a forward/backward mask mismatch or a misplaced
site shows up as a plain numerical difference.  The masks themselves are checked separately: values exactly
{0, 1/(1-p)}, keep rate 1-p within 4 sigma, distinct per site / seed / rank, and (taps) y_train == y_eval * mask
bit for bit at the sites whose input does not depend on another dropout.

Gradient tolerance: rel-L2(HIP, fp64 oracle) <= max(1e-3, 2 x rel-L2(torch fp32 oracle, fp64 oracle)) per tensor,
i.e. north_star's 1e-3 wherever torch's own fp32 backward meets it, and never worse than 2x torch elsewhere.
"""
import math

import numpy as np
import pytest
import torch

from oracle import swapnet_oracle as O
from swapnet_amd import engine
from tests import backends
from tests.test_texture_step import noise_bias as tex_noise_bias, vgg_state_dict
from tests.test_warp_step import noise_bias as warp_noise_bias

BACKENDS = [pytest.param("sim", id="hostsim"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _ctx(kind):
    return backends.gpu_ctx() if kind == "gpu" else backends.hostsim_ctx()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def noise_bias(name, keys=()):
    # biases feeding an InstanceNorm: true gradient 0, the reference's value is round-off
    if name.startswith("unet."):
        return tex_noise_bias(name, keys)
    return warp_noise_bias(name)


def _phased_step(m, labels, training, seed):
    m.forward(training, seed)
    m.backward_D(labels[0], labels[1])
    gD = m.state_dict(engine.NET_D, which=engine.W_GRAD, to_cpu=True)
    m.optimizer_step(engine.NET_D)
    m.backward_G(labels[2])
    gG = m.state_dict(engine.NET_G, which=engine.W_GRAD, to_cpu=True)
    m.optimizer_step(engine.NET_G)
    return gD, gG


def _check_step(m, st, gD, gG, tol_loss=1e-3, tol_out=1e-3, tol_gD=5e-3, tol_gG=1e-2, tol_post=1e-3, what=""):
    L = m.losses()
    for k, v in st.losses.items():
        assert abs(L[k] - v) <= tol_loss * abs(v) + 1e-6, (what, "loss", k, L[k], v)
    assert rel(m.output(), st.fakes) < tol_out, (what, "fakes", rel(m.output(), st.fakes))
    worst = {}
    for name, got, ref, tol in (("gradD", gD, st.grads_D, tol_gD), ("gradG", gG, st.grads_G, tol_gG)):
        for k, v in ref.items():
            if noise_bias(k, list(ref)):
                continue
            e = rel(got[k], v)
            worst[name] = max(worst.get(name, 0.0), e)
            assert e < tol, (what, name, k, e)
    worst.update(_check_post_step(m, st, tol_post, what))
    return worst


def _check_post_step(m, st, tol_post=1e-3, what=""):
    """Post-step weights of both networks against an oracle `st` that has taken the same step (fp32, or float64 with the pattern pinned)."""
    worst = {}
    # post-step weights.  The first AdamW step moves every element by lr * g / (|g| + eps): an element whose gradient
    # is round-off-sized relative to its tensor (|g| < 5e-2 rms(g)) gets a sign-like update of arbitrary sign in the
    # reference too, so those elements are compared on the un-amplified quantity only (their gradient, above) and
    # the weight check covers the rest -- at the full-size shapes a few such elements exist in otherwise clean
    # tensors (e.g. one unit of the innermost U-Net conv bias), unlike the "noise biases" which are noise throughout.
    pG, pD = m.state_dict(engine.NET_G, to_cpu=True), m.state_dict(engine.NET_D, to_cpu=True)
    for name, got, ref, gref in (("postG", pG, st.G, st.grads_G), ("postD", pD, st.D, st.grads_D)):
        for k, v in ref.items():
            if noise_bias(k, list(ref)):
                continue
            g = gref[k].double()
            # "solid" = above what the gradient check itself tolerates: |g| >= 1e-2 rms(g).  Below that the SIGN that Adam's first
            # step turns into +-lr is not determined.  One tensor is known to need more room and is named instead of widening the
            # mask for everybody (round-4 advice): the bias of the innermost U-Net conv (1 x 1 map, 512 units, no norm behind it) --
            # at C3 bs 16 in training mode one unit at 1.3e-2 rms flipped while the gradient tensor agreed to 3.6e-3.
            innermost = k.startswith("unet.") and k.endswith(".model.1.bias") and \
                k.count("model.3") == max(q.count("model.3") for q in ref if q.endswith(".model.1.bias"))
            solid = g.abs() >= (5e-2 if innermost else 1e-2) * g.pow(2).mean().sqrt()
            if not bool(solid.any()):
                continue
            a, b = got[k].double().cpu()[solid], v.double()[solid]
            e = float((a - b).norm() / (b.norm() + 1e-30))
            worst[name] = max(worst.get(name, 0.0), e)
            assert e < tol_post, (what, name, k, e)
    return worst


# ---------------------------------------------------------------------------------------------------------------
# dropout semantics through the C-ABI
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_dropout_mask_semantics(backend):
    ctx = _ctx(backend)
    B, H = 2, 64
    torch.manual_seed(0)
    G = O.warp_module_params()
    batch = O.synth_warp_batch(B, H, H, seed=1234)
    m = backends.get_model(ctx, "warp", B, H)
    backends.reset_state(m, {engine.NET_G: G, engine.NET_D: O.patchgan_params(22)})
    for i, t in enumerate(batch):
        m.set_input(i, t)
    masks = m.dropout_masks(engine.NET_G, seed=7)
    # the reference's 7 sites, in forward order, with the shapes of the layers they follow
    assert [tuple(t.shape) for t, _ in masks] == [(B, 512, H // 16, H // 16), (B, 1024, H // 32, H // 32),
                                                    (B, 1024, H // 64, H // 64)] + [(B, 1024, H // 16, H // 16)] * 4
    for t, p in masks:
        assert p == 0.5
        vals = torch.unique(t.cpu())
        assert vals.tolist() == [0.0, 2.0], vals                      # exactly {0, 1/(1-p)}
        n = t.numel()
        keep = float((t > 0).double().mean())
        assert abs(keep - 0.5) < 4 * 0.5 / math.sqrt(n) + 1e-9, (keep, n)
    cpu = [t.cpu() for t, _ in masks]
    # independent streams: per site (the four resblock sites share a shape), per seed, per rank (base_gan seeds
    # rank r with seed + r, swapnet_amd/models/base_gan.py)
    for i in range(3, 7):
        for j in range(i + 1, 7):
            assert 0.4 < float((cpu[i] == cpu[j]).double().mean()) < 0.6
    other = [t.cpu() for t, _ in m.dropout_masks(engine.NET_G, seed=8)]
    for a, b in zip(cpu, other):
        if a.numel() >= 4096:
            assert 0.4 < float((a == b).double().mean()) < 0.6
    again = [t.cpu() for t, _ in m.dropout_masks(engine.NET_G, seed=7)]
    assert all(torch.equal(a, b) for a, b in zip(cpu, again))
    # forward applies exactly this factor: at the sites fed by dropout-free layers y_train == y_eval * mask
    m.forward(False, 0)
    ev = {k: m.tap(engine.NET_G, k).cpu() for k in ("body_d4", "cloth_d5")}
    m.forward(True, 7)
    tr = {k: m.tap(engine.NET_G, k).cpu() for k in ("body_d4", "cloth_d5", "cloth_d6", "res0_h")}
    assert torch.equal(tr["body_d4"], ev["body_d4"] * cpu[0])
    assert torch.equal(tr["cloth_d5"], ev["cloth_d5"] * cpu[1])
    # downstream sites: zero exactly where the mask is zero
    assert float(tr["cloth_d6"][cpu[2] == 0].abs().max()) == 0.0
    assert float(tr["res0_h"][cpu[3] == 0].abs().max()) == 0.0
    # eval mode never drops
    m.forward(False, 7)
    assert torch.equal(m.tap(engine.NET_G, "body_d4").cpu(), ev["body_d4"])


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_op_level_dropout_forward_backward_share_one_mask(backend):
    """InstanceNorm -> ReLU -> Dropout as one op (ResidualBlock's first half, modules/layers.py:133-136) through
    swn_op_norm_act_dropout: y in {0, 2*relu(IN(x))}, and the input gradient equals autograd's with that mask."""
    ctx = _ctx(backend)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 12, 12, generator=g)
    dy = torch.randn(2, 8, 12, 12, generator=g)
    y, mask, dx = engine.op_norm_act_dropout(ctx, x, dy, act=2, p=0.5, seed=11)
    y, mask, dx = y.cpu(), mask.cpu(), dx.cpu()
    assert torch.unique(mask).tolist() == [0.0, 2.0]
    xr = x.clone().requires_grad_(True)
    ref = torch.relu(torch.nn.functional.instance_norm(xr, eps=1e-5)) * mask
    ref.backward(dy)
    assert rel(y, ref.detach()) < 1e-5
    assert rel(dx, xr.grad) < 1e-4
    assert float(y[mask == 0].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------
# training-mode step == oracle with the replayed masks
# ---------------------------------------------------------------------------------------------------------------
def _warp_train_case(ctx, B, H, seed_w, seed_b, drop_seed, model=None):
    torch.manual_seed(seed_w)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=seed_b)
    m = model or engine.NativeModel(ctx, "warp", B, H, H)
    backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
    for i, t in enumerate(batch):
        m.set_input(i, t)
    masks = [t.cpu() for t, _ in m.dropout_masks(engine.NET_G, seed=drop_seed)]
    return m, G, D, batch, masks


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_warp_training_mode_step_matches_oracle(backend):
    ctx = _ctx(backend)
    labels = [0.9, 0.8, 1.0]
    m, G, D, batch, masks = _warp_train_case(ctx, 2, 64, 0, 1234, 41, model=backends.get_model(ctx, "warp", 2, 64))
    replay = O.MaskReplay(masks)
    st = O.WarpStepOracle(G, D, training=replay)
    st.step(*batch, labels=labels)
    assert replay.done()
    gD, gG = _phased_step(m, labels, True, 41)
    _check_step(m, st, gD, gG, what="warp train 64")
    # and the step really was stochastic: the eval-mode loss differs
    ev = O.WarpStepOracle(G, D)
    ev.step(*batch, labels=labels)
    assert rel(ev.fakes, st.fakes) > 1e-2


# (BASELINE.json C2 exactly -- 256 x 256, bs 32, train mode, bench.py's routing -- is held to the oracle in ONE test since round 6:
# tests/test_pattern_replay.py::test_warp_c2_full_batch_training_step_with_pinned_pattern checks losses, the generated batch, every
# gradient tensor at 1e-4 against float64 with the pattern pinned, the post-step weights and the launch list.  The un-pinned bs-32
# comparison that stood here ran a second CPU oracle step of the same kernels for a 5e-3 / 1e-2 bar: 37-44 s of a 1 200 s budget.)


@pytest.mark.gpu
@pytest.mark.small_channel_winograd
def test_warp_c2_full_batch_step_with_winograd_forms_on_every_level():
    """The harder-numerics variant of the C2 step: SWN_WINO_MINC=32 moves body/cloth_down2 and PatchGAN's model.2
    (64 -> 128 channels at 64x64, the largest-M launches of the direct ring kernel) onto strided Winograd as well.  NOT the
    routing bench.py times (the test above is); kept so that the Winograd forms are held to the oracle at bs 32 on every
    level they can serve (bs 8 since round 5)."""
    ctx = backends.gpu_ctx()
    labels = [0.85, 0.95, 0.75]
    m, G, D, batch, masks = _warp_train_case(ctx, 8, 256, 5, 77, 1234)       # (bs 8: a non-default routing does not need the bs-32 oracle step)
    try:
        st = O.WarpStepOracle(G, D, training=O.MaskReplay(masks))
        st.step(*batch, labels=labels)
        with backends.traced_route(ctx) as route:
            gD, gG = _phased_step(m, labels, True, 1234)
        worst = _check_step(m, st, gD, gG, what="warp 256x256 bs8 train, Winograd everywhere")
        print("warp 256x256 bs8 (SWN_WINO_MINC=32) worst rel-L2:", {k: "%.2e" % v for k, v in worst.items()})
        assert any(l.startswith("body_down2.model.0 f ") and ",b25," in l for l in route.lines), route.lines[:12]
    finally:
        m.close()


def _texture_case(ctx, B, H, drop_seed):
    torch.manual_seed(9)
    G, D = O.texture_module_params(img_size=H), O.patchgan_params(22)
    vgg = O.vgg16_feature_params()
    batch = O.synth_texture_batch(B, H, H, seed=31)
    m = engine.NativeModel(ctx, "texture", B, H, H)
    backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
    m.load_state_dict(engine.NET_VGG, vgg_state_dict(m, vgg))
    for i, t in enumerate(batch):
        m.set_input(i, t)
    masks = [t.cpu() for t, _ in m.dropout_masks(engine.NET_G, seed=drop_seed)]
    return m, G, D, vgg, batch, masks


# (BASELINE.json C3 -- texture 256 x 256, bs 16, 12 ROIs, L1 + VGG16 content + style -- likewise:
# tests/test_pattern_replay.py::test_texture_c3_full_batch_training_step_with_pinned_pattern.)


# ---------------------------------------------------------------------------------------------------------------
# the gradient tolerance, earned: HIP vs an fp64 evaluation, next to torch's own fp32 backward
# ---------------------------------------------------------------------------------------------------------------
def _fp64_yardstick(G, D, batch, labels, training32, training64):
    s32 = O.WarpStepOracle(G, D, training=training32)
    s32.step(*batch, labels=labels)
    s64 = O.WarpStepOracle(G, D, training=training64, dtype=torch.float64)
    s64.step(*batch, labels=labels)
    return s32, s64


def _assert_vs_fp64(got, s32, s64, which, what, floor=1e-3, mult=2.0, cap=None):
    ref64 = getattr(s64, which)
    w = backends.assert_grads_vs_fp64(got, getattr(s32, which), ref64, lambda k: noise_bias(k, list(ref64)), (what, which), floor, mult, cap)
    return [("worst", w[0], w[1])]


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend,winograd", [("sim", "on"), ("sim", "off"), pytest.param("gpu", "on", marks=pytest.mark.gpu)],
                         ids=["on-hostsim", "off-hostsim", "on-mi355x"])     # (Winograd off on the GPU: test_resblock_conv_variants_match_oracle[direct])
def test_gradients_against_fp64_oracle(backend, winograd, monkeypatch):
    """Per-tensor gradient error of the native step measured against the SAME step in float64, beside the error of
    torch's fp32 CPU backward (the reference's arithmetic): HIP <= max(1e-3, 2 x torch-fp32).  Winograd
    F(4x4,3x3) / F(3x3,4x4) on and off (fp32 Winograd is where a looser bound could hide a real defect)."""
    if winograd == "off":
        monkeypatch.setenv("SWN_WINOGRAD", "0")
    ctx = _ctx(backend)
    labels = [0.9, 0.8, 1.0]
    B, H = 2, 64
    torch.manual_seed(0)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=1234)
    s32, s64 = _fp64_yardstick(G, D, batch, labels, False, False)
    m = engine.NativeModel(ctx, "warp", B, H, H)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        gD, gG = _phased_step(m, labels, False, 0)
        rows = _assert_vs_fp64(gD, s32, s64, "grads_D", "64x64") + _assert_vs_fp64(gG, s32, s64, "grads_G", "64x64")
        print("max HIP err vs fp64 %.2e ; max torch-fp32 err vs fp64 %.2e" % (max(r[1] for r in rows), max(r[2] for r in rows)))
        assert rel(m.output(), s64.fakes) < 1e-4
    finally:
        m.close()


@pytest.mark.gpu
def test_gradients_against_fp64_oracle_at_full_resolution():
    """The same yardstick at 256x256 (bs 2): F(4x4,3x3) on the 16x16 maps with 1024-channel reductions, the fused
    tail kernels, F(3x3,4x4) on PatchGAN's 31x31 map.  Un-pinned, so bounded by the sign-flip floor (5e-3, see
    backends.assert_grads_vs_fp64); the same case with the activation pattern pinned is held to 1e-4 in
    tests/test_pattern_replay.py."""
    ctx = backends.gpu_ctx()
    labels = [0.9, 0.8, 1.0]
    B, H = 2, 256
    torch.manual_seed(3)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=99)
    s32, s64 = _fp64_yardstick(G, D, batch, labels, False, False)
    m = engine.NativeModel(ctx, "warp", B, H, H)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        gD, gG = _phased_step(m, labels, False, 0)
        rows = _assert_vs_fp64(gD, s32, s64, "grads_D", "256x256", 1e-3, 4.0, 5e-3) + _assert_vs_fp64(gG, s32, s64, "grads_G", "256x256", 1e-3, 4.0, 5e-3)
        print("max HIP err vs fp64 %.2e ; max torch-fp32 err vs fp64 %.2e" % (max(r[1] for r in rows), max(r[2] for r in rows)))
    finally:
        m.close()


@pytest.mark.gpu
def test_warp_c2_eval_mode_forward_matches_oracle_at_the_benchmarked_batch():
    """Eval-mode routing at BASELINE.json C2's batch (256 x 256, bs 32: the un-split 36-plane resblock launches, the bs-32 split-K plans of
    the deep cloth levels) held to the oracle's FORWARD pass (round-5 advice: the full-batch oracle STEPS of the suite are train-mode
    only since round 5; a forward-only oracle pass is ~10 s of CPU).  Generated batch and CE term 1e-3 (north_star), observed ~1e-6; the
    eval-mode forward launches are a subset of the default-environment training step's forward launches."""
    ctx = backends.gpu_ctx()
    B, H = 32, 256
    torch.manual_seed(3)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=99)
    m = backends.get_model(ctx, "warp", B, H)
    backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
    for i, t in enumerate(batch):
        m.set_input(i, t)
    with backends.traced_route(ctx) as route:
        m.forward(False, 0)
    got = m.output().cpu()
    with torch.no_grad():
        ref = O.warp_module_forward(G, batch[0], batch[1], training=False)
    err = rel(got, ref)
    print("warp 256x256 bs32 eval forward: fakes rel-L2 %.2e" % err)
    assert err < 1e-3, err
    ce_got = float(torch.nn.functional.cross_entropy(got, torch.argmax(batch[2], dim=1)))
    ce_ref = float(torch.nn.functional.cross_entropy(ref, torch.argmax(batch[2], dim=1)))
    assert abs(ce_got - ce_ref) <= 1e-3 * abs(ce_ref), (ce_got, ce_ref)
    fwd = [l for l in route.lines if " f " in l]
    assert any(",b36,full1152," in l for l in fwd), fwd[:10]                     # the un-split 36-plane launch of the resblocks
    want = [l for l in backends.default_route("warp", B, H) if " f " in l and not l.startswith("model.")]
    extra = [l for l in fwd if l not in want]
    assert not extra, ("eval-mode forward launched what the default training step's forward does not", extra[:8])


@pytest.mark.gpu
def test_training_mode_loss_statistics_match_oracle():
    """Train-mode statistics without replaying masks: the mean generator loss over K independent dropout draws
    (library RNG) against the oracle's mean over K draws of torch's RNG -- same distribution, different streams."""
    ctx = backends.gpu_ctx()
    B, H, K = 4, 64, 24
    torch.manual_seed(0)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=5)
    m = engine.NativeModel(ctx, "warp", B, H, H)
    try:
        got, ref = [], []
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        w0 = [m.weight_arena(net).clone() for net in (engine.NET_G, engine.NET_D)]      # packed weights, on the device
        lambda_ce = O.WarpStepOracle(G, D).h["lambda_ce"]
        for k in range(K):
            # every draw from the same state: weights restored device-to-device (weight_arena() marks them changed: the derived
            # operands are refreshed), both Adam moments and the step counters zeroed -- not 140 M parameters uploaded per draw
            for net, w in zip((engine.NET_G, engine.NET_D), w0):
                m.weight_arena(net).copy_(w)
                m.arena(net, engine.W_EXP_AVG).zero_(); m.arena(net, engine.W_EXP_AVG_SQ).zero_()
                m.optim_step_count(net, 0)
            m.ctx.sync()
            m.step([0.9, 0.8, 1.0], training=True, seed=1000 + k)
            got.append(m.losses()["G_ce"])
            # the oracle's G_ce needs the train-mode FORWARD only (warp_model.py:141-147: lambda_ce x CE of the generated batch): a whole
            # optimize_parameters of the 140 M-parameter port per draw was 3-7 s of CPU on the GPU box, 70-100 s of the suite
            torch.manual_seed(2000 + k)
            with torch.no_grad():
                fakes = O.warp_module_forward(G, batch[0], batch[1], training=True)
                ref.append(float(torch.nn.functional.cross_entropy(fakes, torch.argmax(batch[2], dim=1))) * lambda_ce)
        got, ref = np.array(got), np.array(ref)
        se = math.sqrt(got.var(ddof=1) / K + ref.var(ddof=1) / K)
        assert abs(got.mean() - ref.mean()) < 4 * se + 1e-3 * abs(ref.mean()), (got.mean(), ref.mean(), se)
        assert 0.5 < got.std(ddof=1) / ref.std(ddof=1) < 2.0
    finally:
        m.close()


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_destroying_a_model_returns_its_memory(backend):
    """A model owns every device buffer it allocates -- at construction and lazily (private PatchGAN of
    discriminate(), gradient-penalty operands, perceptual scratch) -- and swn_model_destroy hands them back: a process
    that builds models for several shapes (NativeBackend's per-shape cache, the two-stage pipeline) does not
    accumulate HBM.  Handles may be destroyed in any order (the context lives until its last model)."""
    ctx = _ctx(backend)
    base = ctx.bytes_allocated()
    sizes = []
    for kind in ("warp", "texture", "warp"):
        m = engine.NativeModel(ctx, kind, 2, 64, 64)
        if kind == "warp":
            m.discriminate(torch.zeros(2, 22, 64, 64))        # lazily built private PatchGAN
        sizes.append(ctx.bytes_allocated() - base)
        assert sizes[-1] > (1 << 20)
        m.close()
        assert ctx.bytes_allocated() == base
    assert sizes[0] == sizes[2]
    # a context handle destroyed before its model: the model still works and still frees
    c2 = engine.Context(workspace_mb=64) if backend == "gpu" else backends.hostsim_ctx(fresh=True)
    m = engine.NativeModel(c2, "warp", 1, 64, 64)
    c2.close()
    m.close()


@pytest.mark.gpu
def test_conv_epilogue_instance_norm_statistics_match_the_statistics_pass(monkeypatch):
    """north_star's Conv + InstanceNorm fusion (modules/layers.py:12-24): above 1024 pixels per plane the 128 x 128 ring
    kernel's epilogue leaves the statistics' partial sums and norm_act_fwd skips its statistics pass -- where the launch runs whole tiles
    anyway (bs 16 at 256 x 256: body_down2 / cloth_down2 are 512 tiles).  The same forward with SWN_CONV_STATS=0 (statistics pass) must give
    the same normalised levels to fp32 round-off of the two statistics (fp64 sums either way), the route must say which form ran, and a batch
    the planner would split along K (bs 2) must NOT take the fused form (it would trade short accumulation chains for one long one)."""
    ctx = backends.gpu_ctx()
    torch.manual_seed(4)
    G = O.warp_module_params()
    taps, routes = {}, {}
    for B in (16, 2):
        batch = O.synth_warp_batch(B, 256, 256, seed=31)
        for stats in ("1", "0"):
            monkeypatch.setenv("SWN_CONV_STATS", stats)
            m = engine.NativeModel(ctx, "warp", B, 256, 256, is_train=False)
            try:
                m.load_state_dict(engine.NET_G, G)
                m.set_input(0, batch[0]); m.set_input(1, batch[1])
                with backends.traced_route(ctx) as r:
                    m.forward(False, 0)
                routes[B, stats] = [l for l in r.lines if "norm_act[" in l]
                taps[B, stats] = {k: m.tap(engine.NET_G, k).cpu() for k in ("body_d2", "cloth_d2", "body_d3", "dual_u3")}
            finally:
                m.close()
    fused = [l for l in routes[16, "1"] if "conv epilogue" in l]
    assert len(fused) >= 1, routes[16, "1"]          # body_down2, cloth_down2 (the report lists distinct (label, phase, kernel) triples)
    assert not [l for l in routes[16, "0"] if "conv epilogue" in l] and not [l for l in routes[2, "1"] if "conv epilogue" in l]
    for k in taps[16, "1"]:
        e = rel(taps[16, "1"][k], taps[16, "0"][k])
        assert e < 2e-6, (k, e)
    assert all(torch.equal(taps[2, "1"][k], taps[2, "0"][k]) for k in taps[2, "1"])        # bs 2: the same kernels either way
