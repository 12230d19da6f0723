"""The reference's representation options (options/base_options.py:75-105): --body_representation labels with
--body_channels 12 (neural-body-fitting segmentations), --cloth_representation rgb (3 channels), other
--cloth_channels.  WarpModel resolves them to the channel counts of WarpModule and of the conditional PatchGAN
(models/warp_model.py:49-55,89,97); TextureModel concatenates --cloth_channels cloth channels behind the 36 ROI-pooled
ones (models/texture_model.py:94-109).  A full native step against the oracle for each combination."""
import pytest
import torch

from oracle import swapnet_oracle as O
from swapnet_amd import engine
from tests import backends
from tests.test_texture_step import vgg_state_dict
from tests.test_train_parity import BACKENDS, _ctx, _phased_step, noise_bias, rel

pytestmark = pytest.mark.small_channel_winograd      # tests/conftest.py: small shapes on the Winograd forms


def _blocky_onehot(B, H, C, g):
    lab = torch.randint(0, C, (B, H // 8, H // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
    return lab, O.labels_to_onehot(lab, C).contiguous()


# (the channel-option matrix is outside SURVEY 8's rows: on the MI355X one case per test -- the reference-golden test below runs all
# three representations there -- the full matrix on the host simulator)
@pytest.mark.parametrize("backend,channels", [("sim", (12, 19)), ("sim", (3, 3)), ("sim", (12, 7)),
                                              pytest.param("gpu", (12, 7), marks=pytest.mark.gpu)],
                         ids=["body-labels-hostsim", "cloth-rgb-hostsim", "body12-cloth7-hostsim", "body12-cloth7-mi355x"])
def test_warp_step_with_other_representations(backend, channels):
    Cb, Cc = channels
    ctx = _ctx(backend)
    B, H = 2, 64
    torch.manual_seed(0)
    G, D = O.warp_module_params(Cb, Cc), O.patchgan_params(Cb + Cc)
    g = torch.Generator().manual_seed(5)
    bodys = torch.randn(B, Cb, H, H, generator=g)
    lab, targets = _blocky_onehot(B, H, Cc, g)
    inputs = O.labels_to_onehot(torch.roll(lab.flip(2), (3, -2), (1, 2)), Cc).contiguous()
    labels = [0.9, 0.8, 1.0]
    m = engine.NativeModel(ctx, "warp", B, H, H, body_channels=Cb, cloth_channels=Cc)
    try:
        assert m.param_infos(engine.NET_G)["cloth_down1.model.0.weight"] == (64, Cc, 4, 4)
        assert m.param_infos(engine.NET_D)["model.0.weight"] == (64, Cb + Cc, 4, 4)
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        m.set_input(0, bodys); m.set_input(1, inputs)
        m.set_input_labels(2, lab.to(torch.int32))              # device-side one-hot with Cc classes
        gD, gG = _phased_step(m, labels, False, 0)
        # float64 oracle with the native pass's activation pattern pinned (tests/test_pattern_replay.py)
        st = O.WarpStepOracle(G, D, dtype=torch.float64)
        st.patterns = O.PatternReplay(backends.collect_patterns(m))
        st.step(bodys, inputs, targets, labels=labels)
        st.patterns.check()
        L = m.losses()
        for k, v in st.losses.items():
            assert abs(L[k] - v) <= 2e-5 * abs(v) + 1e-6, (k, L[k], v)
        assert tuple(m.output().shape) == (B, Cc, H, H) and rel(m.output(), st.fakes) < 2e-5
        backends.assert_grads_replayed(gD, st.grads_D, noise_bias, 1e-4, ("D", Cb, Cc))
        backends.assert_grads_replayed(gG, st.grads_G, noise_bias, 1e-4, ("G", Cb, Cc))
        with pytest.raises(ValueError):
            m.set_input(1, torch.zeros(B, Cc + 1, H, H))
        pred = m.discriminate(torch.cat((bodys, targets), 1))
        assert rel(pred, O.patchgan_forward(st.D, torch.cat((bodys, targets), 1).double())) < 1e-3      # the updated D
    finally:
        m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_texture_step_with_other_cloth_channels(backend):
    ctx = _ctx(backend)
    B, H, Cc = 2, 64, 7
    torch.manual_seed(9)
    G, D = O.texture_module_params(cloth_channels=Cc, img_size=H), O.patchgan_params(3 + Cc)
    vgg = O.vgg16_feature_params()
    tex, rois, _, tgt = O.synth_texture_batch(B, H, H, seed=31)
    _, cloths = _blocky_onehot(B, H, Cc, torch.Generator().manual_seed(2))
    labels = [0.85, 0.95, 0.75]
    m = engine.NativeModel(ctx, "texture", B, H, H, cloth_channels=Cc)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        m.load_state_dict(engine.NET_VGG, vgg_state_dict(m, vgg))
        for i, t in enumerate((tex, rois, cloths, tgt)):
            m.set_input(i, t)
        gD, gG = _phased_step(m, labels, False, 0)
        st = O.TextureStepOracle(G, D, vgg, dtype=torch.float64)
        st.patterns = O.PatternReplay(backends.collect_patterns(m, vgg=True))
        st.step(tex, rois, cloths, tgt, labels=labels)
        st.patterns.check()
        L = m.losses()
        for k, v in st.losses.items():
            assert abs(L[k] - v) <= 2e-5 * abs(v) + 1e-6, (k, L[k], v)
        assert rel(m.output(), st.fakes) < 2e-5
        backends.assert_grads_replayed(gD, st.grads_D, lambda k: noise_bias(k, list(st.grads_D)), 1e-4, ("tex D", Cc))
        backends.assert_grads_replayed(gG, st.grads_G, lambda k: noise_bias(k, list(st.grads_G)), 1e-4, ("tex G", Cc))
    finally:
        m.close()


def test_warp_model_resolves_the_representation_flags(tmp_path):
    """models/warp_model.py:49-55: body channels = --body_channels under --body_representation labels (else 3), cloth
    channels = 3 under --cloth_representation rgb (else --cloth_channels); the generator, the conditional PatchGAN and
    one optimize_parameters() step follow."""
    from swapnet_amd.models import create_model
    from tests.test_models_api import make_opt
    for kw, (cb, cc) in ((dict(body_representation="labels", body_channels=12), (12, 19)),
                         (dict(cloth_representation="rgb"), (3, 3)),
                         (dict(cloth_channels=7), (3, 7))):
        opt = make_opt(tmp_path, "sim", **kw)
        torch.manual_seed(0)
        model = create_model(opt)
        model.setup(opt)
        sdG, sdD = model.net_generator.state_dict(), model.net_discriminator.state_dict()
        assert tuple(sdG["body_down1.model.0.weight"].shape) == (64, cb, 4, 4)
        assert tuple(sdG["cloth_down1.model.0.weight"].shape) == (64, cc, 4, 4)
        assert tuple(sdG["upsample_and_pad.2.weight"].shape) == (cc, 192, 4, 4)
        assert tuple(sdD["model.0.weight"].shape) == (64, cb + cc, 4, 4) and model.get_D_inchannels() == cb + cc
        g = torch.Generator().manual_seed(1)
        lab, targets = _blocky_onehot(2, 64, cc, g)
        model.set_input(dict(bodys=torch.randn(2, cb, 64, 64, generator=g), input_cloths=targets.flip(3).contiguous(),
                             target_cloths=targets, cloth_paths=["", ""], body_paths=["", ""]))
        model.optimize_parameters()
        losses = model.get_current_losses()
        assert all(v == v and abs(v) < 1e6 for v in losses.values()), losses
        assert tuple(model.fakes.shape) == (2, cc, 64, 64)


@pytest.mark.parametrize("backend", BACKENDS)
def test_channel_options_reproduce_the_reference(backend, golden_dir):
    """tests/golden/warp_channels_64.npz was recorded from the REAL reference (oracle/make_golden.py channels:
    WarpModel under --body_representation labels / --cloth_representation rgb / --cloth_channels 7).  The oracle's
    seeded construction reproduces its initial weights, and the oracle step and the native step reproduce its losses,
    fakes and post-step weights."""
    import os
    import numpy as np
    from oracle.golden_io import compare
    from oracle.golden_io import sample_idx
    gold = np.load(os.path.join(golden_dir, "warp_channels_64.npz"))
    B, H = int(gold["meta/B"]), int(gold["meta/H"])

    def post_ok(key, t, lr):
        """Post-step weights: norm to 1e-3 and the 24 samples to 1e-3 -- except that AdamW's first update is
        lr * g / (|g| + eps), a sign-like +-lr for an element whose gradient is round-off sized: up to two samples may
        differ by that 2 lr (the reference's own value is as arbitrary as ours there)."""
        t = t.detach().double().cpu().reshape(-1)
        gn, gs = float(gold[key + "/norm"]), np.asarray(gold[key + "/samples"], dtype=np.float64)
        mine = t[torch.from_numpy(sample_idx(t.numel(), key))].numpy()
        err = np.abs(mine - gs)
        tol = 1e-3 * np.abs(gs) + 3e-3 * gn / np.sqrt(t.numel())
        out = err > tol
        return abs(float(t.norm()) - gn) <= 1e-3 * gn and out.sum() <= 2 and bool(np.all(err[out] <= 2.2 * lr)), (key, err.max())
    ctx = _ctx(backend)
    for tag in ("body-labels", "cloth-rgb", "cloth7"):
        cb, cc = (int(v) for v in gold[tag + "/meta/channels"])
        pre = tag + "/"
        torch.manual_seed(int(gold["meta/init_seed"]))
        G, D = O.warp_module_params(cb, cc), O.patchgan_params(cb + cc)
        for k in ("body_down1.model.0.weight", "cloth_down1.model.0.weight", "upsample_and_pad.2.weight"):
            ok, msg = compare(gold, pre + "init/G/" + k, G[k], 1e-6, 1e-6)
            assert ok, msg
        ok, msg = compare(gold, pre + "init/D/model.0.weight", D["model.0.weight"], 1e-6, 1e-6)
        assert ok, msg
        batch = O.synth_channels_batch(B, H, cb, cc)
        torch.manual_seed(int(gold["meta/step_seed"]))
        st = O.WarpStepOracle(G, D)
        st.step(*batch)                                     # labels drawn from the CPU RNG in the reference's order
        m = engine.NativeModel(ctx, "warp", B, H, H, body_channels=cb, cloth_channels=cc)
        try:
            backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
            for i, t in enumerate(batch):
                m.set_input(i, t)
            m.step(st.labels, training=False, seed=0)
            L = m.losses()
            for k, v in st.losses.items():
                ref = float(gold[pre + "loss/" + k])
                assert abs(v - ref) <= 1e-4 * abs(ref) + 1e-6, (tag, "oracle", k, v, ref)
                assert abs(L[k] - ref) <= 1e-3 * abs(ref) + 1e-6, (tag, "native", k, L[k], ref)
            for who, fakes in (("oracle", st.fakes), ("native", m.output())):
                ok, msg = compare(gold, pre + "fakes", fakes, 1e-3, 1e-3)
                assert ok, (who, msg)
            pG, pD = m.state_dict(engine.NET_G, to_cpu=True), m.state_dict(engine.NET_D, to_cpu=True)
            for k in ("body_down1.model.0.weight", "cloth_down1.model.0.weight", "upsample_and_pad.2.weight",
                      "resblocks.3.conv_block.6.weight"):
                for who, sd in (("oracle", st.G), ("native", pG)):
                    ok, msg = post_ok(pre + "postG/" + k, sd[k], 1e-4)
                    assert ok, (who, msg)
            for who, sd in (("oracle", st.D), ("native", pD)):
                ok, msg = post_ok(pre + "postD/model.0.weight", sd["model.0.weight"], 4e-4)
                assert ok, (who, msg)
        finally:
            m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_patchgan_depth_option_reproduces_the_reference(backend, golden_dir):
    """--discriminator n_layers --n_layers_D 2 / 4 (modules/discriminators.py:45-88,91-136; models/base_gan.py:147-149).
    tests/golden/warp_depths_64.npz was recorded from the REAL reference (oracle/make_golden.py depths): parameter set after
    init, the discriminator's prediction map on the conditioned targets, one WarpModel step.  The oracle reproduces all of it;
    the native PatchGAN (swn_ctx_set_patchgan_layers) has the reference's parameter names / shapes, its standalone forward
    matches, and its fused step reproduces losses, fakes and every post-step D weight."""
    import os
    import numpy as np
    from oracle.golden_io import compare
    gold = np.load(os.path.join(golden_dir, "warp_depths_64.npz"))
    B, H = int(gold["meta/B"]), int(gold["meta/H"])
    ctx = _ctx(backend)
    bodys, inputs, targets = O.synth_warp_batch(B, H, H, seed=1234)
    for n in (2, 4):
        pre = "n%d/" % n
        torch.manual_seed(int(gold["meta/init_seed"]))
        G, D = O.warp_module_params(), O.patchgan_params(22, n_layers=n)
        assert list(D.keys()) == [str(k) for k in gold[pre + "D_keys"]]
        for k, v in D.items():
            assert tuple(v.shape) == tuple(int(x) for x in gold[pre + "D_shape/" + k])
            ok, msg = compare(gold, pre + "initD/" + k, v, 1e-6, 1e-6)
            assert ok, msg
        x = torch.cat((bodys, targets), 1)
        with torch.no_grad():
            pred = O.patchgan_forward(D, x)
        ok, msg = compare(gold, pre + "pred_real", pred, 1e-4, 1e-4)
        assert ok, ("oracle", msg)
        torch.manual_seed(int(gold["meta/step_seed"]))
        st = O.WarpStepOracle(G, D)
        st.step(bodys, inputs, targets)
        m = engine.NativeModel(ctx, "warp", B, H, H, n_layers_D=n)
        try:
            infos = m.param_infos(engine.NET_D)
            assert list(infos.keys()) == list(D.keys()) and all(tuple(infos[k]) == tuple(D[k].shape) for k in D)
            backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
            got = m.discriminate(x)
            assert tuple(got.shape) == tuple(pred.shape) and rel(got, pred) < 1e-3, (n, rel(got, pred))
            ok, msg = compare(gold, pre + "pred_real", got, 1e-3, 1e-3)
            assert ok, ("native", msg)
            for i, t in enumerate((bodys, inputs, targets)):
                m.set_input(i, t)
            m.step(st.labels, training=False, seed=0)
            L = m.losses()
            for k, v in st.losses.items():
                ref = float(gold[pre + "loss/" + k])
                assert abs(v - ref) <= 1e-4 * abs(ref) + 1e-6, (n, "oracle", k, v, ref)
                assert abs(L[k] - ref) <= 1e-3 * abs(ref) + 1e-6, (n, "native", k, L[k], ref)
            for who, fakes in (("oracle", st.fakes), ("native", m.output())):
                ok, msg = compare(gold, pre + "fakes", fakes, 1e-3, 1e-3)
                assert ok, (who, msg)
            pD = m.state_dict(engine.NET_D, to_cpu=True)
            for k in D:
                if k.endswith(".weight"):
                    assert rel(pD[k], st.D[k]) < 1e-3, (n, k, rel(pD[k], st.D[k]))
                    ok, msg = compare(gold, pre + "postD/" + k, st.D[k], 1e-3, 3e-3)
                    assert ok, ("oracle", msg)
            # (the gradient-penalty objectives at these depths: tests/test_gradient_penalty.py::test_gradient_penalty_at_other_patchgan_depths)
        finally:
            m.close()
    with pytest.raises(ValueError):
        engine.NativeModel(ctx, "warp", B, H, H, n_layers_D=6)
    with pytest.raises(ValueError):
        engine.NativeModel(ctx, "warp", 1, 32, 32, n_layers_D=4)          # 32 >> 4 = 2 pixels: too small
