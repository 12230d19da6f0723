"""--discriminator pixel (models/base_gan.py:61-65 -> modules/discriminators.py:39-41,139-175): the third choice of the reference's
discriminator option, a 1x1 "pixelGAN" -- Conv1x1(22, 64) - LeakyReLU - Conv1x1(64, 128) - InstanceNorm - LeakyReLU - Conv1x1(128, 1), a
real / fake prediction per pixel.  tests/golden/warp_pixel_64.npz was recorded from the REAL reference (oracle/make_golden.py pixel):
parameter set after init, the prediction map on the conditioned targets, one WarpModel step.  Held to it: the oracle, the native
network (PatchGAN "depth 0" of a context, conv kind k1s1) through the C-ABI, and the drop-in model API (create_model(opt))."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import swapnet_oracle as O
from oracle.golden_io import compare
from swapnet_amd import engine
from tests import backends
from tests.test_models_api import make_opt
from tests.test_ops import rel, run_conv
from tests.test_train_parity import _ctx

SIM = pytest.param("sim", id="hostsim")
GPU = pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)


@pytest.mark.parametrize("backend", [SIM])        # (SURVEY 2 #4 marks PixelDiscriminator OUT OF SCOPE: one GPU case -- the reference golden below -- is kept)
def test_one_by_one_conv_operator(backend):
    """swn_op_conv kind 5: forward, input gradient, weight gradient of a 1x1 stride-1 conv against float64 -- the first layer's 22 -> 64
    (padded input buffer), the 64 -> 128 middle, the 1-channel head, and a ring-kernel-sized case."""
    ctx = _ctx(backend)
    g = torch.Generator().manual_seed(1)
    for n, ci, co, h in ((2, 22, 64, 16), (1, 64, 128, 12), (2, 128, 1, 10), (2, 64, 128, 64)):
        x = torch.randn(n, ci, h, h, generator=g)
        w = torch.randn(co, ci, 1, 1, generator=g) * (2.0 / ci) ** 0.5
        b = torch.randn(co, generator=g) * 0.1
        ref = F.conv2d(x.double(), w.double(), b.double())
        out = run_conv(ctx, 5, 0, 0, False, x, w, b, 0, ref.shape)
        dy = torch.randn(ref.shape, generator=g)
        xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
        F.conv2d(xr, wr).backward(dy.double())
        dx = run_conv(ctx, 5, 0, 2, False, torch.zeros_like(x), w, None, 0, dy=dy)
        dw = run_conv(ctx, 5, 0, 1, False, x, torch.zeros_like(w), None, 0, dy=dy)
        for what, a, r in (("fwd", out, ref), ("dgrad", dx, xr.grad), ("wgrad", dw, wr.grad)):
            assert rel(a, r) < 1e-5, (ci, co, h, what, rel(a, r))


def test_oracle_pixel_discriminator_matches_reference(golden_dir):
    gold = np.load(os.path.join(golden_dir, "warp_pixel_64.npz"))
    B, H = int(gold["meta/B"]), int(gold["meta/H"])
    torch.manual_seed(int(gold["meta/init_seed"]))
    G, D = O.warp_module_params(), O.pixelgan_params(22)
    assert list(D.keys()) == [str(k) for k in gold["pixel/D_keys"]] == ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.5.weight", "net.5.bias"]
    for k, v in D.items():
        assert tuple(v.shape) == tuple(int(x) for x in gold["pixel/D_shape/" + k])
        ok, msg = compare(gold, "pixel/initD/" + k, v, 1e-6, 1e-6)
        assert ok, msg
    bodys, inputs, targets = O.synth_warp_batch(B, H, H, seed=1234)
    with torch.no_grad():
        pred = O.patchgan_forward(D, torch.cat((bodys, targets), 1))
    assert tuple(pred.shape) == (B, 1, H, H)
    ok, msg = compare(gold, "pixel/pred_real", pred, 1e-4, 1e-4)
    assert ok, msg
    torch.manual_seed(int(gold["meta/step_seed"]))
    st = O.WarpStepOracle(G, D)
    losses = st.step(bodys, inputs, targets)
    for k, v in losses.items():
        ref = float(gold["pixel/loss/" + k])
        assert abs(v - ref) <= 1e-4 * abs(ref) + 1e-6, (k, v, ref)
    ok, msg = compare(gold, "pixel/fakes", st.fakes, 1e-4, 1e-4)
    assert ok, msg
    for k in D:
        ok, msg = compare(gold, "pixel/postD/" + k, st.D[k], 1e-3, 3e-3)
        assert ok, msg
    for k in ("body_down1.model.0.weight", "upsample_and_pad.2.weight"):
        ok, msg = compare(gold, "pixel/postG/" + k, st.G[k], 1e-3, 3e-3)
        assert ok, msg


@pytest.mark.parametrize("backend", [SIM, GPU])
def test_native_pixel_discriminator_reproduces_the_reference(backend, golden_dir):
    gold = np.load(os.path.join(golden_dir, "warp_pixel_64.npz"))
    B, H = int(gold["meta/B"]), int(gold["meta/H"])
    ctx = _ctx(backend)
    torch.manual_seed(int(gold["meta/init_seed"]))
    G, D = O.warp_module_params(), O.pixelgan_params(22)
    bodys, inputs, targets = O.synth_warp_batch(B, H, H, seed=1234)
    x = torch.cat((bodys, targets), 1)
    torch.manual_seed(int(gold["meta/step_seed"]))
    st = O.WarpStepOracle(G, D)
    st.step(bodys, inputs, targets)
    m = engine.NativeModel(ctx, "warp", B, H, H, n_layers_D=0)
    try:
        infos = m.param_infos(engine.NET_D)
        assert list(infos.keys()) == list(D.keys()) and all(tuple(infos[k]) == tuple(D[k].shape) for k in D)
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        got = m.discriminate(x)
        assert tuple(got.shape) == (B, 1, H, H)
        ok, msg = compare(gold, "pixel/pred_real", got, 1e-3, 1e-3)
        assert ok, msg
        for i, t in enumerate((bodys, inputs, targets)):
            m.set_input(i, t)
        # phase by phase (models/base_gan.py:194-203), every gradient of both networks against the oracle's
        m.forward(False, 0)
        m.backward_D(st.labels[0], st.labels[1])
        gD = m.state_dict(engine.NET_D, which=engine.W_GRAD, to_cpu=True)
        m.optimizer_step(engine.NET_D)
        m.backward_G(st.labels[2])
        gG = m.state_dict(engine.NET_G, which=engine.W_GRAD, to_cpu=True)
        m.optimizer_step(engine.NET_G)
        for k, v in st.grads_D.items():
            if k != "net.2.bias":          # feeds the InstanceNorm: its true gradient is 0, both sides hold round-off
                assert rel(gD[k], v) < 1e-3, ("gradD", k, rel(gD[k], v))
        worst = max(rel(gG[k], v) for k, v in st.grads_G.items() if k.endswith(".weight"))
        assert worst < 5e-3, ("gradG", worst)
        L = m.losses()
        for k in st.losses:
            ref = float(gold["pixel/loss/" + k])
            assert abs(L[k] - ref) <= 1e-3 * abs(ref) + 1e-6, (k, L[k], ref)
        ok, msg = compare(gold, "pixel/fakes", m.output(), 1e-3, 1e-3)
        assert ok, msg
        pD = m.state_dict(engine.NET_D, to_cpu=True)
        for k in D:
            if k.endswith(".weight"):
                ok, msg = compare(gold, "pixel/postD/" + k, pD[k], 1e-3, 3e-3)
                assert ok, msg
        # the fused step is the same step
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        m.step(st.labels, training=False, seed=0)
        pD2 = m.state_dict(engine.NET_D, to_cpu=True)
        assert all(torch.equal(pD[k], pD2[k]) for k in pD)
        with pytest.raises(Exception):
            m.set_hyper(gp_mode=1, gan_mode=2)            # the gradient penalty walks the NLayerDiscriminator only
    finally:
        m.close()


def test_model_api_with_the_pixel_discriminator(tmp_path, golden_dir):
    """create_model(opt) with --discriminator pixel, seeded like the golden run: loss dict and checkpoint keys of the reference."""
    from swapnet_amd.models import create_model
    gold = np.load(os.path.join(golden_dir, "warp_pixel_64.npz"))
    model = create_model(make_opt(tmp_path, "sim", discriminator="pixel"))
    assert list(model.net_discriminator.state_dict().keys()) == [str(k) for k in gold["pixel/D_keys"]]
    torch.manual_seed(int(gold["meta/init_seed"]))
    model.net_generator.load_state_dict(O.warp_module_params())
    model.net_discriminator.load_state_dict(O.pixelgan_params(22))
    model.eval()
    bodys, inputs, targets = O.synth_warp_batch(2, 64, 64, seed=1234)
    model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets, cloth_paths=["", ""], body_paths=["", ""]))
    torch.manual_seed(int(gold["meta/step_seed"]))
    model.optimize_parameters()
    for k, v in model.get_current_losses().items():
        ref = float(gold["pixel/loss/" + k])
        assert abs(v - ref) <= 1e-3 * abs(ref) + 1e-6, (k, v, ref)
    ok, msg = compare(gold, "pixel/postD/net.2.weight", model.net_discriminator.state_dict()["net.2.weight"], 1e-3, 3e-3)
    assert ok, msg
    pred = model.net_discriminator(torch.cat((bodys, targets), 1))
    assert tuple(pred.shape) == (2, 1, 64, 64)
    with pytest.raises(NotImplementedError):
        create_model(make_opt(tmp_path, "sim", discriminator="pixel", gan_mode="wgan-gp"))


@pytest.mark.parametrize("backend", [SIM])
def test_texture_stage_against_the_pixel_discriminator(backend):
    """The texture stage builds its discriminator through the same factory (models/base_gan.py:147-149): one step against the oracle."""
    from tests.test_texture_step import vgg_state_dict
    from tests.test_train_parity import _check_step, _phased_step
    ctx = _ctx(backend)
    B, H = 1, 64
    torch.manual_seed(4)
    G, D, vgg = O.texture_module_params(img_size=H), O.pixelgan_params(22), O.vgg16_feature_params()
    batch = O.synth_texture_batch(B, H, H, seed=12)
    m = engine.NativeModel(ctx, "texture", B, H, H, n_layers_D=0)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        m.load_state_dict(engine.NET_VGG, vgg_state_dict(m, vgg))
        for i, t in enumerate(batch):
            m.set_input(i, t)
        labels = [0.85, 0.95, 0.75]
        st = O.TextureStepOracle(G, D, vgg)
        st.step(*batch, labels=labels)
        gD, gG = _phased_step(m, labels, False, 0)
        gD = {k: v for k, v in gD.items()}
        st.grads_D.pop("net.2.bias", None); st.D.pop("net.2.bias", None)          # (feeds the InstanceNorm: round-off on both sides)
        _check_step(m, st, gD, gG, what="texture step, pixel discriminator")
    finally:
        m.close()
