"""Gradient parity with the activation pattern pinned.

Two fp32 evaluations of these networks in different summation orders (this library's MFMA tiles vs torch's CPU
kernels) leave a handful of LeakyReLU / ReLU pre-activations -- and max-pool windows -- within round-off of a tie on
opposite sides.  Each flip changes d(loss)/dx of that element by O(1); on PatchGAN at 256x256, 2-6 flips out of 2M
elements are the ENTIRE ~1e-3 rel-L2 distance of any fp32 gradient (torch's own fp32 backward included) from the
float64 one (tools/d_grad_trace.py).  These tests take the branch pattern the native pass
actually used (swn_model_act_pattern), replay it in the float64 oracle's backward, and compare every gradient tensor
at a tolerance ten times tighter than the 1e-3 of the un-pinned comparison (observed: D <= 5e-6, G <= 2e-5 at bs 2,
6e-5 on one inner U-Net tensor at bs 16) -- in eval and (with the dropout masks replayed as well) in training mode."""
import pytest
import torch

from oracle import swapnet_oracle as O
from swapnet_amd import engine
from tests import backends
from tests.test_train_parity import BACKENDS, _check_post_step, _ctx, _phased_step, _texture_case, noise_bias, rel

TOL = 1e-4


def _warp_replay(ctx, B, H, seed, training, labels=(0.9, 0.8, 1.0), drop_seed=77, check_route=False, tol=TOL, tol_fwd=2e-5,
                 flip_fraction=1e-3):
    torch.manual_seed(seed)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=99)
    m = engine.NativeModel(ctx, "warp", B, H, H)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        masks = [t.cpu() for t, _ in m.dropout_masks(engine.NET_G, seed=drop_seed)] if training else None
        with backends.traced_route(ctx) as route:
            gD, gG = _phased_step(m, list(labels), training, drop_seed)
        if check_route:            # the benchmarked configurations: bench.py's kernel routing, launch for launch
            backends.assert_default_routing(route.lines, "warp", B, H)
        replay = O.PatternReplay(backends.collect_patterns(m))
        s64 = O.WarpStepOracle(G, D, training=O.MaskReplay(masks) if training else False, dtype=torch.float64)
        s64.patterns = replay
        s64.step(*batch, labels=list(labels))
        flips = replay.check(max_fraction=flip_fraction)
        wD = backends.assert_grads_replayed(gD, s64.grads_D, lambda k: noise_bias(k, list(s64.grads_D)), tol, ("warp", H, "D"))
        wG = backends.assert_grads_replayed(gG, s64.grads_G, lambda k: noise_bias(k, list(s64.grads_G)), tol, ("warp", H, "G"))
        assert rel(m.output(), s64.fakes) < tol_fwd, rel(m.output(), s64.fakes)
        L = m.losses()
        for k, v in s64.losses.items():
            assert abs(L[k] - v) <= tol_fwd * abs(v) + 1e-7, (k, L[k], v)
        _warp_replay.last_forward_error = rel(m.output(), s64.fakes)
        if check_route:            # the full-batch case also holds the post-step weights (AdamW on the pinned float64 gradients) to 1e-3
            print("post-step weights", {k: "%.1e" % v for k, v in _check_post_step(m, s64, 1e-3, ("warp", H, "post")).items()})
        return flips, wD, wG
    finally:
        m.close()


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_warp_gradients_with_pinned_pattern(backend, mode):
    flips, wD, wG = _warp_replay(_ctx(backend), 2, 64, 0, mode == "train")
    print("warp 64x64", mode, "flips", flips, "worst D %.2e G %.2e" % (wD, wG))


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("kind", ["warp", "texture"])
def test_gradients_with_pinned_pattern_under_the_simulated_16_bit_operand_formats(kind, monkeypatch, capfd):
    """"dtype f32" rests on 22-bit operands: every ring-kernel GEMM multiplies two fp16 planes h + l of x 2^k per operand, k from
    the tensor's amax -- taken from the slot its producer filled, from a bound (pair-form Winograd planes: the transform's gain x
    the input's slot) or handed over from a layer's other operand (weights).  The device kernels are not runnable here, but their
    operand ROUNDING is: with SWN_SIM_PAIR=1 the host simulator cuts every operand the way the device stores or loads it (software
    fp16: pre-cut weights and pair-form planes round h, an activation cut in the loop truncates it, dY in the weight gradient
    rounds it) with the scales the ENGINE chooses, and multiplies the results.  So the engine's whole scale plumbing and the
    precision it leaves run in CPU CI: pinned gradients must stay within the same 1e-4 of float64 (measured here: warp D 2.1e-6 /
    G 1.4e-5, texture 3.6e-6; on the MI355X: 5e-6 / 7e-6 ... 1.2e-5, 5e-5)."""
    monkeypatch.setenv("SWN_SIM_PAIR", "1")
    monkeypatch.setenv("SWN_SIM_SLOT_REPORT", "1e30")          # ([pair] lines on stderr: proof that planes were rounded)
    if kind == "warp":
        flips, wD, wG = _warp_replay(backends.hostsim_ctx(), 2, 64, 0, True)
    else:
        flips, wD, wG = _texture_replay(backends.hostsim_ctx(), 2, 64, False)
    err = capfd.readouterr().err
    if kind == "warp":
        assert "[pair] wino_input_transform" in err and "[pair] wino_dy_transform" in err and "[pair] wino_s2_input_transform" in err, err[-400:]
    print(kind, "64x64, simulated 16-bit operand formats: flips", flips, "worst D %.2e G %.2e" % (wD, wG))


def _texture_replay(ctx, B, H, training, labels=(0.85, 0.95, 0.75), drop_seed=99, check_route=False):
    m, G, D, vgg, batch, masks = _texture_case(ctx, B, H, drop_seed)
    try:
        with backends.traced_route(ctx) as route:
            gD, gG = _phased_step(m, list(labels), training, drop_seed)
        if check_route:
            backends.assert_default_routing(route.lines, "texture", B, H)
        replay = O.PatternReplay(backends.collect_patterns(m, vgg=True))
        s64 = O.TextureStepOracle(G, D, vgg, training=O.MaskReplay(masks) if training else False, dtype=torch.float64)
        s64.patterns = replay
        s64.step(*batch, labels=list(labels))
        flips = replay.check()
        wD = backends.assert_grads_replayed(gD, s64.grads_D, lambda k: noise_bias(k, list(s64.grads_D)), TOL, ("texture", H, "D"))
        wG = backends.assert_grads_replayed(gG, s64.grads_G, lambda k: noise_bias(k, list(s64.grads_G)), TOL, ("texture", H, "G"))
        assert rel(m.output(), s64.fakes) < 2e-5
        L = m.losses()
        for k, v in s64.losses.items():
            assert abs(L[k] - v) <= 1e-4 * abs(v) + 1e-7, (k, L[k], v)
        if check_route:
            print("post-step weights", {k: "%.1e" % v for k, v in _check_post_step(m, s64, 1e-3, ("texture", H, "post")).items()})
        return flips, wD, wG
    finally:
        m.close()


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_texture_gradients_with_pinned_pattern(backend, mode):
    flips, wD, wG = _texture_replay(_ctx(backend), 2, 64, mode == "train")
    print("texture 64x64", mode, "flips", flips, "worst D %.2e G %.2e" % (wD, wG))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_warp_gradients_with_pinned_pattern_at_full_resolution(mode):
    """256x256: every kernel family of the benchmarked configuration (LDS-DMA ring tiles, Winograd F(4x4,3x3) /
    F(3x3,4x4), the fused tail kernels, hybrid split-K) with the comparison no longer dominated by sign flips."""
    flips, wD, wG = _warp_replay(backends.gpu_ctx(), 2, 256, 3, mode == "train")
    print("warp 256x256", mode, "flips", flips, "worst D %.2e G %.2e" % (wD, wG))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_texture_gradients_with_pinned_pattern_at_full_resolution(mode):
    flips, wD, wG = _texture_replay(backends.gpu_ctx(), 2, 256, mode == "train")
    print("texture 256x256", mode, "flips", flips, "worst D %.2e G %.2e" % (wD, wG))


@pytest.mark.gpu
def test_warp_c2_full_batch_training_step_with_pinned_pattern():
    """BASELINE.json C2 exactly as bench.py times it (256x256, bs 32, TRAINING mode, default kernel routing -- the launch list
    is compared with a scrubbed-environment process): dropout masks and activation pattern replayed in the float64 oracle,
    losses and the generated batch within 2e-5, every gradient tensor within 1e-4, the post-step weights within 1e-3."""
    flips, wD, wG = _warp_replay(backends.gpu_ctx(), 32, 256, 3, True, check_route=True)
    print("warp C2 bs32 train", "flips", flips, "worst D %.2e G %.2e" % (wD, wG))


@pytest.mark.gpu
def test_texture_c3_full_batch_training_step_with_pinned_pattern():
    """BASELINE.json C3 (256x256, bs 16, 12 ROIs, L1 + VGG16 content + style, TRAINING mode), same comparison."""
    flips, wD, wG = _texture_replay(backends.gpu_ctx(), 16, 256, True, check_route=True)
    print("texture C3 bs16 train", "flips", flips, "worst D %.2e G %.2e" % (wD, wG))


def test_unpinned_gradient_distance_is_branch_flips():
    """CPU-only demonstration of the claim the tolerances rest on: the ~1e-3 rel-L2 distance of an
    fp32 gradient from the float64 one at 256x256 is NOT arithmetic error.  The oracle evaluates the discriminator step
    of the warp stage in fp32 and in float64: un-pinned, PatchGAN's weight gradients are 1e-4 .. 2e-3 apart; with the
    fp32 run's LeakyReLU branches replayed in the float64 run (a handful of elements out of millions differ) the
    distance collapses by two to three orders of magnitude."""
    B, H = 2, 256
    torch.manual_seed(3)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    bodys, inputs, targets = O.synth_warp_batch(B, H, H, seed=99)
    with torch.no_grad():
        fakes = O.warp_module_forward(G, bodys, inputs)
    x = torch.cat((torch.cat((bodys, fakes), 1), torch.cat((bodys, targets), 1)), 0)      # [fake | real], conditioned

    def d_step(dtype, patterns):
        P = {k: v.to(dtype).requires_grad_(True) for k, v in D.items()}
        with patterns.scope("D"):
            pred = O.patchgan_forward(P, x.to(dtype))
        loss = 0.5 * (O.gan_loss(pred[:B], torch.tensor([0.9], dtype=dtype)) + O.gan_loss(pred[B:], torch.tensor([0.8], dtype=dtype)))
        grads = torch.autograd.grad(loss, list(P.values()))
        return dict(zip(P.keys(), grads))

    rec32, rec64 = O.PatternReplay(), O.PatternReplay()
    g32 = d_step(torch.float32, rec32)
    g64 = d_step(torch.float64, rec64)                                    # float64 on its own branches
    pinned = O.PatternReplay(rec32.groups)
    g64p = d_step(torch.float64, pinned)                                  # float64 on the fp32 run's branches
    flips = pinned.check()["D"]
    keys = [k for k in g64 if k.endswith("weight") and k != "model.11.weight"]
    un = max(rel(g32[k], g64[k]) for k in keys)
    pn = max(rel(g32[k], g64p[k]) for k in keys)
    print("flips %d of %d elements; un-pinned %.2e, pinned %.2e" % (flips, pinned.elements["D"], un, pn))
    assert pn < 2e-5
    if flips:                      # (a seed without a single flip would make both distances round-off sized)
        assert un > 20 * pn and flips < 1e-4 * pinned.elements["D"]


# ---- the reduced-precision configuration (BASELINE.json C4's arithmetic on one GPU): one fp16 plane per operand --------------
_F16_SCRIPT = r"""
import sys
sys.path.insert(0, %(repo)r)
from tests import backends
from tests.test_pattern_replay import _warp_replay
flips, wD, wG = _warp_replay(backends.gpu_ctx(), 2, 256, 3, True, tol=1.0, tol_fwd=1.0, flip_fraction=0.05)
print("F16RESULT %%g %%g %%g" %% (wD, wG, _warp_replay.last_forward_error), flips)
"""


@pytest.mark.gpu
def test_one_plane_configuration_tolerance_study():
    """`bench.py --precision f16` (SWN_PC_PLANES=1, SWN_WGRAD_PLANES=1): every ring-kernel GEMM multiplies ONE fp16 plane of
    each amax-scaled operand (11 mantissa bits; one MFMA per product instead of three), storage and accumulation stay fp32.
    Not the parity configuration -- north_star's 1e-3 belongs to the two-plane form above -- but held to a measured bound with
    the same instrument: activation pattern and dropout masks replayed in the float64 oracle, 256x256, training mode.  The plane
    count is read once per process, so the step runs in a subprocess."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, SWN_PC_PLANES="1", SWN_WGRAD_PLANES="1")
    out = subprocess.run([sys.executable, "-c", _F16_SCRIPT % dict(repo=backends.REPO)], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("F16RESULT")][-1]
    wD, wG, fwd = (float(v) for v in line.split()[1:4])
    print("one fp16 plane per operand, warp 256x256 train: worst pinned-pattern gradient error D %.2e G %.2e, output %.2e" % (wD, wG, fwd))
    # measured on MI355X (round 4): output 3.1e-3, gradients D 6.5e-3 / G 8.8e-3 of the float64 values with the pattern pinned
    # -- the arithmetic of 11-bit operands through ~25 layers; bounds at three times that
    assert fwd < 1e-2 and wD < 2e-2 and wG < 3e-2, (wD, wG, fwd)
