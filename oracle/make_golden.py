"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REAL
reference (/root/reference, imported through oracle/ref_stubs.py) on CPU.

    python -m oracle.make_golden            # from the repo root, build container only

The reference holds no golden vectors of its own, so these
files are the pin for oracle/swapnet_oracle.py (tests/test_oracle_golden.py) and,
through it, for the HIP path.  Because a WarpModule is 137.6 M parameters the
fixtures do not store tensors: weights are reproduced from the seed on both
sides (same torch build in the container and on the GPU box) and every tensor
is recorded as (L2 norm, sum, 24 sampled elements at fixed flat indices).
"""
import argparse
import os
import sys
import tempfile
from collections import OrderedDict

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ref_stubs  # noqa: E402

from oracle.golden_io import summarize as _summarize, store_full, FULL_TENSORS  # noqa: E402

_FULL = set()
_FULL_STRIDE = [1]


def summarize(out, key, t):
    _summarize(out, key, t)
    if key in _FULL:
        store_full(out, key, t, stride=_FULL_STRIDE[0] if key.endswith("/fakes") else 1)


def base_opt(tmp, **kw):
    o = dict(
        gpu_id=None, is_train=True, checkpoints_dir=tmp, name="golden", no_confirm=True,
        body_channels=12, body_representation="rgb", cloth_channels=19,
        cloth_representation="labels", texture_channels=3, init_type="kaiming",
        init_gain=0.02, discriminator="basic", n_layers_D=3, norm="instance",
        gan_label_mode="smooth", gan_mode="vanilla", lambda_discriminator=1.0,
        lambda_gan=1.0, lambda_gp=10, optimizer_G="AdamW", optimizer_D="AdamW",
        lr=1e-4, d_lr=4e-4, weight_decay=0.0, d_weight_decay=0.01, b1=0.9, b2=0.999,
        beta1=0.5, verbose=False, continue_train=False, load_epoch="latest",
        body_norm_stats=((0.0,) * 3, (1.0,) * 3), texture_norm_stats=((0.0,) * 3, (1.0,) * 3),
    )
    o.update(kw)
    return argparse.Namespace(**o)


def hook_taps(net, names):
    taps, handles = OrderedDict(), []
    for n in names:
        mod = dict(net.named_modules())[n]
        handles.append(mod.register_forward_hook(
            lambda m, i, o, n=n: taps.__setitem__(n, o.detach().clone())))
    return taps, handles


def golden_cloth_format(path_npz, path_expected, H=24, W=16, n_labels=19, seed=7):
    """A cloth segmentation written by the reference's compress_and_save_cloth and what its
    decompress_cloth_segment reads back (datasets/data_utils.py:298-343)."""
    from datasets import data_utils as R
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, n_labels, (H, W), generator=g)
    lab[:, :3] = 0                                       # background columns (not stored)
    scores = torch.rand((n_labels, H, W), generator=g) * 0.5
    scores.scatter_(0, lab[None], 1.0)                   # argmax == lab
    R.compress_and_save_cloth(scores, path_npz)
    onehot = R.decompress_cloth_segment(path_npz, n_labels)
    np.savez(path_expected, labels=lab.numpy().astype(np.int32), scores=scores.numpy(),
             onehot=onehot.numpy(), n_labels=np.int64(n_labels))
    print("wrote", path_npz, path_expected)


def golden_roi_ops(path, seed=11):
    """crop_rois / flip_rois_ of the reference (datasets/data_utils.py:197-295) on seeded ROI tables."""
    from datasets import data_utils as R
    g = torch.Generator().manual_seed(seed)
    xy = torch.randint(0, 200, (12, 2), generator=g)
    wh = torch.randint(0, 90, (12, 2), generator=g)
    rois = torch.cat([xy, xy + wh], 1).float()
    rois[3] = rois[3, [0, 1, 0, 1]]                       # degenerate box
    bounds = ((40, 32), (168, 224))
    out = {"rois": rois.numpy(), "bounds": np.array(bounds),
           "crop_torch": R.crop_rois(rois, bounds).numpy(),
           "crop_numpy": R.crop_rois(rois.numpy().astype(np.int64), bounds)}
    for axis, center in ((0, 128), (1, 96)):
        r = rois.clone()
        R.flip_rois_(r, axis, center)
        out["flip%d" % axis] = r.numpy()
        out["center%d" % axis] = np.int64(center)
    np.savez(path, **out)
    print("wrote", path)


def golden_warp(path, H=64, B=2, init_seed=0, step_seeds=(100, 101), full="warp"):
    """Steps of the REAL WarpModel (models/warp_model.py:106-183 under models/base_gan.py:194-203).  full = which FULL_TENSORS set is
    stored whole: "warp" (64 x 64, bs 2), "warp_256" (BASELINE.json C2's resolution: 256 x 256, the resblocks on 16 x 16 maps, PatchGAN on
    31 x 31 -- fakes at stride 4), "warp_c1" (C1: 64 x 64, bs 4)."""
    from oracle.swapnet_oracle import synth_warp_batch
    from models.warp_model import WarpModel
    from oracle.golden_io import FULL_STRIDE
    out = OrderedDict()
    _FULL.clear(); _FULL.update(FULL_TENSORS[full]); _FULL_STRIDE[0] = FULL_STRIDE.get(full, 1)
    with tempfile.TemporaryDirectory() as tmp:
        opt = base_opt(tmp, warp_mode="gan", lambda_ce=100.0, model="warp")
        torch.manual_seed(init_seed)
        model = WarpModel(opt)
        model.eval()                      # dropout off; IN has no running stats
        for k, v in model.net_generator.state_dict().items():
            summarize(out, "init/G/" + k, v)
        for k, v in model.net_discriminator.state_dict().items():
            summarize(out, "init/D/" + k, v)
        bodys, inputs, targets = synth_warp_batch(B, H, H, seed=1234)
        tap_names = ["body_down1", "body_down2", "body_down3", "body_down4", "cloth_down1",
                     "cloth_down2", "cloth_down3", "cloth_down4", "cloth_down5", "cloth_down6",
                     "cloth_up1", "cloth_up2", "resblocks.0", "resblocks.1", "resblocks.2",
                     "resblocks.3", "dual_up1", "dual_up2", "dual_up3"]
        taps, handles = hook_taps(model.net_generator, tap_names)
        for si, seed in enumerate(step_seeds):
            model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets,
                                 cloth_paths=[""] * B, body_paths=[""] * B))
            torch.manual_seed(seed)
            # record the three smooth-label draws the step is about to make
            st = torch.get_rng_state()
            lows = [float(torch.rand(1) * (torch.tensor(1.1) - torch.tensor(0.7)) + torch.tensor(0.7))
                    for _ in range(3)]
            torch.set_rng_state(st)
            # capture D grads before optimizer_G.zero_grad is irrelevant: D grads are
            # overwritten by backward_G (quirk 5), so snapshot them via a hook on step
            d_grads = {}
            orig_step = model.optimizer_D.step

            def step_and_snap(*a, **k):
                for n, p in model.net_discriminator.named_parameters():
                    d_grads[n] = p.grad.detach().clone()
                return orig_step(*a, **k)
            model.optimizer_D.step = step_and_snap
            model.optimize_parameters()
            model.optimizer_D.step = orig_step
            pre = "step%d/" % si
            out[pre + "labels"] = np.array(lows, dtype=np.float64)
            for k, v in model.get_current_losses().items():
                out[pre + "loss/" + k] = np.float64(v)
            summarize(out, pre + "fakes", model.fakes)
            if si == 0:
                for n, t in taps.items():
                    summarize(out, "fwd/" + n, t)
            for n, p in model.net_generator.named_parameters():
                summarize(out, pre + "gradG/" + n, p.grad)
                summarize(out, pre + "postG/" + n, p)
            for n, p in model.net_discriminator.named_parameters():
                summarize(out, pre + "gradD/" + n, d_grads[n])
                summarize(out, pre + "postD/" + n, p)
        for h in handles:
            h.remove()
        # integer work: label decode of the generated batch (util/decode_labels.py)
        dec = model.__class__.compute_visuals
        from util.decode_labels import decode_cloth_labels
        out["decode/fakes_rgb"] = decode_cloth_labels(model.fakes[:1, :, :16, :16]).numpy()
        out["decode/argmax"] = model.fakes[:1, :, :16, :16].argmax(dim=1).numpy()
    out["meta/H"] = np.int64(H)
    out["meta/B"] = np.int64(B)
    out["meta/init_seed"] = np.int64(init_seed)
    out["meta/step_seeds"] = np.array(step_seeds, dtype=np.int64)
    out["meta/torch"] = np.array(torch.__version__)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries")


def golden_warp_modes(path, H=64, B=2, init_seed=0, step_seed=100):
    """One reference step of WarpModel under --gan_mode lsgan / wgan and --warp_mode ce
    (modules/loss.py:55-62,117-128; models/warp_model.py:169-183): losses, fakes and two post-step
    weights.  Same init seed / batch / step seed as golden_warp, so the weights and labels coincide."""
    from oracle.swapnet_oracle import synth_warp_batch
    from models.warp_model import WarpModel
    out = OrderedDict()
    bodys, inputs, targets = synth_warp_batch(B, H, H, seed=1234)
    for mode, opts in (("lsgan", dict(gan_mode="lsgan", warp_mode="gan")),
                       ("wgan", dict(gan_mode="wgan", warp_mode="gan")),
                       ("ce", dict(gan_mode="vanilla", warp_mode="ce")),
                       # gradient-penalty objectives (modules/loss.py:133-184: double backward through D)
                       ("wgan-gp", dict(gan_mode="wgan-gp", warp_mode="gan")),
                       ("dragan-gp", dict(gan_mode="dragan-gp", warp_mode="gan")),
                       ("dragan-lp", dict(gan_mode="dragan-lp", warp_mode="gan"))):
        with tempfile.TemporaryDirectory() as tmp:
            opt = base_opt(tmp, lambda_ce=100.0, model="warp", **opts)
            torch.manual_seed(init_seed)
            model = WarpModel(opt)
            model.eval()
            model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets,
                                 cloth_paths=[""] * B, body_paths=[""] * B))
            torch.manual_seed(step_seed)
            model.optimize_parameters()
            pre = mode + "/"
            for k, v in model.get_current_losses().items():
                out[pre + "loss/" + k] = np.float64(v)
            summarize(out, pre + "fakes", model.fakes)
            sd = model.net_generator.state_dict()
            for k in ("upsample_and_pad.2.weight", "resblocks.3.conv_block.6.weight", "body_down1.model.0.weight"):
                summarize(out, pre + "postG/" + k, sd[k])
            if hasattr(model, "net_discriminator"):
                dsd = model.net_discriminator.state_dict()
                summarize(out, pre + "postD/model.0.weight", dsd["model.0.weight"])
                if "p" in mode[-2:]:
                    for k in ("model.5.weight", "model.8.weight", "model.11.weight"):
                        summarize(out, pre + "postD/" + k, dsd[k])
    out["meta/init_seed"] = np.int64(init_seed); out["meta/step_seed"] = np.int64(step_seed)
    out["meta/B"] = np.int64(B); out["meta/H"] = np.int64(H)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries")


def golden_warp_nonsquare(path, H=128, W=64, B=1, init_seed=0, step_seed=100):
    """One step of the REAL WarpModel on a 2:1 batch (BASELINE.json C5 is DeepFashion's 4:3, 256 x 192: the warp generator is fully
    convolutional -- modules/swapnet_modules.py:92-151 -- and the conditional PatchGAN -- modules/discriminators.py:111-131 -- takes any
    map): losses, fakes, post-step weights of both networks.  Pins the oracle (and through it the library) at H != W."""
    from oracle.swapnet_oracle import synth_warp_batch
    from models.warp_model import WarpModel
    out = OrderedDict()
    bodys, inputs, targets = synth_warp_batch(B, H, W, seed=1234)
    with tempfile.TemporaryDirectory() as tmp:
        opt = base_opt(tmp, lambda_ce=100.0, model="warp", warp_mode="gan")
        torch.manual_seed(init_seed)
        model = WarpModel(opt)
        model.eval()
        model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets, cloth_paths=[""] * B, body_paths=[""] * B))
        torch.manual_seed(step_seed)
        model.optimize_parameters()
        for k, v in model.get_current_losses().items():
            out["loss/" + k] = np.float64(v)
        assert tuple(model.fakes.shape) == (B, 19, H, W)
        summarize(out, "fakes", model.fakes)
        sd = model.net_generator.state_dict()
        for k in ("upsample_and_pad.2.weight", "resblocks.3.conv_block.6.weight", "body_down1.model.0.weight", "cloth_down6.model.0.weight"):
            summarize(out, "postG/" + k, sd[k])
        dsd = model.net_discriminator.state_dict()
        for k in ("model.0.weight", "model.8.weight", "model.11.weight"):
            summarize(out, "postD/" + k, dsd[k])
    out["meta/init_seed"] = np.int64(init_seed); out["meta/step_seed"] = np.int64(step_seed)
    out["meta/B"] = np.int64(B); out["meta/H"] = np.int64(H); out["meta/W"] = np.int64(W)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries")


def golden_warp_depths_gp(path, H=64, B=2, init_seed=0, step_seed=100):
    """--gan_mode wgan-gp / dragan-gp with --discriminator n_layers --n_layers_D 2 / 4 (modules/loss.py:133-184 through
    modules/discriminators.py:91-136 at other depths): one step of the REAL reference's WarpModel -- losses, fakes, every
    post-step D weight (they carry the second-order gradient of the penalty)."""
    from oracle.swapnet_oracle import synth_warp_batch
    from models.warp_model import WarpModel
    out = OrderedDict()
    bodys, inputs, targets = synth_warp_batch(B, H, H, seed=1234)
    for n in (2, 4):
        for mode in ("wgan-gp", "dragan-gp"):
            with tempfile.TemporaryDirectory() as tmp:
                opt = base_opt(tmp, lambda_ce=100.0, model="warp", gan_mode=mode, warp_mode="gan", discriminator="n_layers", n_layers_D=n)
                torch.manual_seed(init_seed)
                model = WarpModel(opt)
                model.eval()
                model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets,
                                     cloth_paths=[""] * B, body_paths=[""] * B))
                torch.manual_seed(step_seed)
                model.optimize_parameters()
                pre = "n%d/%s/" % (n, mode)
                for k, v in model.get_current_losses().items():
                    out[pre + "loss/" + k] = np.float64(v)
                summarize(out, pre + "fakes", model.fakes)
                for k, v in model.net_discriminator.state_dict().items():
                    if k.endswith(".weight"):
                        summarize(out, pre + "postD/" + k, v)
    out["meta/init_seed"] = np.int64(init_seed); out["meta/step_seed"] = np.int64(step_seed)
    out["meta/B"] = np.int64(B); out["meta/H"] = np.int64(H)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries")


def golden_warp_depths(path, H=64, B=2, init_seed=0, step_seed=100):
    """--discriminator n_layers --n_layers_D 2 / 4 (modules/discriminators.py:45-88,91-136; models/base_gan.py:147-149): the
    REAL reference's discriminator at other depths -- its parameter names / shapes after init, its prediction map on the
    conditioned target batch, and one WarpModel step (losses, fakes, every post-step D weight)."""
    from oracle.swapnet_oracle import synth_warp_batch
    from models.warp_model import WarpModel
    out = OrderedDict()
    bodys, inputs, targets = synth_warp_batch(B, H, H, seed=1234)
    for n in (2, 4):
        with tempfile.TemporaryDirectory() as tmp:
            opt = base_opt(tmp, lambda_ce=100.0, model="warp", gan_mode="vanilla", warp_mode="gan", discriminator="n_layers", n_layers_D=n)
            torch.manual_seed(init_seed)
            model = WarpModel(opt)
            model.eval()
            pre = "n%d/" % n
            dsd = model.net_discriminator.state_dict()
            out[pre + "D_keys"] = np.array(list(dsd.keys()))
            for k, v in dsd.items():
                out[pre + "D_shape/" + k] = np.array(v.shape, dtype=np.int64)
                summarize(out, pre + "initD/" + k, v)
            with torch.no_grad():
                summarize(out, pre + "pred_real", model.net_discriminator(torch.cat((bodys, targets), 1)))
            model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets,
                                 cloth_paths=[""] * B, body_paths=[""] * B))
            torch.manual_seed(step_seed)
            model.optimize_parameters()
            for k, v in model.get_current_losses().items():
                out[pre + "loss/" + k] = np.float64(v)
            summarize(out, pre + "fakes", model.fakes)
            for k, v in model.net_discriminator.state_dict().items():
                summarize(out, pre + "postD/" + k, v)
            summarize(out, pre + "postG/body_down1.model.0.weight", model.net_generator.state_dict()["body_down1.model.0.weight"])
    out["meta/init_seed"] = np.int64(init_seed); out["meta/step_seed"] = np.int64(step_seed)
    out["meta/B"] = np.int64(B); out["meta/H"] = np.int64(H)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries")


def golden_warp_pixel(path, H=64, B=2, init_seed=0, step_seed=100):
    """--discriminator pixel (models/base_gan.py:61-65 -> modules/discriminators.py:39-41,139-175): the REAL reference's
    PixelDiscriminator -- parameter names / shapes after init, its per-pixel prediction map on the conditioned target batch, and one
    WarpModel step against it (losses, fakes, every post-step D weight, one G weight)."""
    from oracle.swapnet_oracle import synth_warp_batch
    from models.warp_model import WarpModel
    out = OrderedDict()
    bodys, inputs, targets = synth_warp_batch(B, H, H, seed=1234)
    with tempfile.TemporaryDirectory() as tmp:
        opt = base_opt(tmp, lambda_ce=100.0, model="warp", gan_mode="vanilla", warp_mode="gan", discriminator="pixel")
        torch.manual_seed(init_seed)
        model = WarpModel(opt)
        model.eval()
        pre = "pixel/"
        dsd = model.net_discriminator.state_dict()
        out[pre + "D_keys"] = np.array(list(dsd.keys()))
        for k, v in dsd.items():
            out[pre + "D_shape/" + k] = np.array(v.shape, dtype=np.int64)
            summarize(out, pre + "initD/" + k, v)
        with torch.no_grad():
            summarize(out, pre + "pred_real", model.net_discriminator(torch.cat((bodys, targets), 1)))
        model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets, cloth_paths=[""] * B, body_paths=[""] * B))
        torch.manual_seed(step_seed)
        model.optimize_parameters()
        for k, v in model.get_current_losses().items():
            out[pre + "loss/" + k] = np.float64(v)
        summarize(out, pre + "fakes", model.fakes)
        for k, v in model.net_discriminator.state_dict().items():
            summarize(out, pre + "postD/" + k, v)
        for k in ("body_down1.model.0.weight", "upsample_and_pad.2.weight"):
            summarize(out, pre + "postG/" + k, model.net_generator.state_dict()[k])
    out["meta/init_seed"] = np.int64(init_seed); out["meta/step_seed"] = np.int64(step_seed)
    out["meta/B"] = np.int64(B); out["meta/H"] = np.int64(H)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries")


def golden_warp_channels(path, H=64, B=2, init_seed=0, step_seed=100):
    """One reference step of WarpModel under the representation options (options/base_options.py:75-105,
    models/warp_model.py:49-55): --body_representation labels (12 body channels), --cloth_representation rgb (3 cloth
    channels), --cloth_channels 7.  Init weights of the first / last generator convs and of PatchGAN's first conv (the
    layers whose shape depends on the options), losses, fakes, post-step weights."""
    from models.warp_model import WarpModel
    from oracle.swapnet_oracle import synth_channels_batch
    out = OrderedDict()
    for tag, opts, (cb, cc) in (("body-labels", dict(body_representation="labels", body_channels=12), (12, 19)),
                                ("cloth-rgb", dict(cloth_representation="rgb"), (3, 3)),
                                ("cloth7", dict(cloth_channels=7), (3, 7))):
        with tempfile.TemporaryDirectory() as tmp:
            opt = base_opt(tmp, lambda_ce=100.0, model="warp", warp_mode="gan", **opts)
            torch.manual_seed(init_seed)
            model = WarpModel(opt)
            model.eval()
            pre = tag + "/"
            sdG, sdD = model.net_generator.state_dict(), model.net_discriminator.state_dict()
            for k in ("body_down1.model.0.weight", "cloth_down1.model.0.weight", "upsample_and_pad.2.weight"):
                summarize(out, pre + "init/G/" + k, sdG[k])
            summarize(out, pre + "init/D/model.0.weight", sdD["model.0.weight"])
            bodys, inputs, targets = synth_channels_batch(B, H, cb, cc)
            model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets,
                                 cloth_paths=[""] * B, body_paths=[""] * B))
            torch.manual_seed(step_seed)
            model.optimize_parameters()
            for k, v in model.get_current_losses().items():
                out[pre + "loss/" + k] = np.float64(v)
            summarize(out, pre + "fakes", model.fakes)
            sdG, sdD = model.net_generator.state_dict(), model.net_discriminator.state_dict()
            for k in ("body_down1.model.0.weight", "cloth_down1.model.0.weight", "upsample_and_pad.2.weight",
                      "resblocks.3.conv_block.6.weight"):
                summarize(out, pre + "postG/" + k, sdG[k])
            summarize(out, pre + "postD/model.0.weight", sdD["model.0.weight"])
            out[pre + "meta/channels"] = np.array([cb, cc], dtype=np.int64)
    out["meta/init_seed"] = np.int64(init_seed); out["meta/step_seed"] = np.int64(step_seed)
    out["meta/B"] = np.int64(B); out["meta/H"] = np.int64(H)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries")


def golden_texture(path, H=64, B=2, init_seed=1, step_seeds=(200, 201), full="texture"):
    """Steps of the REAL TextureModel (models/texture_model.py:121-180).  full="texture_256": BASELINE.json C3's resolution -- the
    U-Net's depth follows the image size (modules/swapnet_modules.py:178: num_downs = frexp(img_size)[1] - 1 = 8 at 256, three
    dropout-carrying inner blocks, modules/pix2pix_modules.py:147-154), RoIAlign 256 -> 128, VGG16 at 256 x 256."""
    from oracle.swapnet_oracle import synth_texture_batch
    from models.texture_model import TextureModel
    from oracle.golden_io import FULL_STRIDE
    out = OrderedDict()
    _FULL.clear(); _FULL.update(FULL_TENSORS[full]); _FULL_STRIDE[0] = FULL_STRIDE.get(full, 1)
    with tempfile.TemporaryDirectory() as tmp:
        opt = base_opt(tmp, model="texture", netG="swapnet", crop_size=H, lambda_l1=10.0,
                       lambda_content=20.0, lambda_style=1e-8)
        torch.manual_seed(init_seed)
        model = TextureModel(opt)
        model.eval()
        for k, v in model.net_generator.state_dict().items():
            summarize(out, "init/G/" + k, v)
        for k, v in model.net_discriminator.state_dict().items():
            summarize(out, "init/D/" + k, v)
        tex, rois, cloths, tgt = synth_texture_batch(B, H, H, seed=4321)
        taps, handles = hook_taps(model.net_generator, ["roi_align", "encode"])
        for si, seed in enumerate(step_seeds):
            model.set_input(dict(input_textures=tex, rois=rois, cloths=cloths, target_textures=tgt,
                                 cloth_paths=[""] * B, texture_paths=[""] * B))
            torch.manual_seed(seed)
            st = torch.get_rng_state()
            lows = [float(torch.rand(1) * (torch.tensor(1.1) - torch.tensor(0.7)) + torch.tensor(0.7))
                    for _ in range(3)]
            torch.set_rng_state(st)
            d_grads = {}
            orig_step = model.optimizer_D.step

            def step_and_snap(*a, **k):
                for n, p in model.net_discriminator.named_parameters():
                    d_grads[n] = p.grad.detach().clone()
                return orig_step(*a, **k)
            model.optimizer_D.step = step_and_snap
            model.optimize_parameters()
            model.optimizer_D.step = orig_step
            pre = "step%d/" % si
            out[pre + "labels"] = np.array(lows, dtype=np.float64)
            for k, v in model.get_current_losses().items():
                out[pre + "loss/" + k] = np.float64(v)
            summarize(out, pre + "fakes", model.fakes)
            if si == 0:
                for n, t in taps.items():
                    summarize(out, "fwd/" + n, t)
            for n, p in model.net_generator.named_parameters():
                summarize(out, pre + "gradG/" + n, p.grad)
                summarize(out, pre + "postG/" + n, p)
            for n, p in model.net_discriminator.named_parameters():
                summarize(out, pre + "gradD/" + n, d_grads[n])
                summarize(out, pre + "postD/" + n, p)
        for h in handles:
            h.remove()
    out["meta/H"] = np.int64(H)
    out["meta/B"] = np.int64(B)
    out["meta/init_seed"] = np.int64(init_seed)
    out["meta/step_seeds"] = np.array(step_seeds, dtype=np.int64)
    out["meta/torch"] = np.array(torch.__version__)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "entries")


def golden_roi(path):
    """The only recorded real-data value in the reference: the (4,12,4) ROI tensor
    printed in `test/Test TextureDataset Draw ROIs.ipynb` cell 9 -- stored so the
    RoIAlign kernel is exercised on realistic (incl. degenerate) boxes."""
    import json
    nb = json.load(open(os.path.join(ref_stubs.REFERENCE_ROOT, "test",
                                     "Test TextureDataset Draw ROIs.ipynb")))
    text = None
    for cell in nb["cells"]:
        for o in cell.get("outputs", []):
            t = "".join(o.get("text", []) or o.get("data", {}).get("text/plain", []))
            if "tensor([[[" in t and text is None:
                text = t
    assert text is not None, "ROI tensor printout not found in the notebook"
    nums = [float(x) for x in __import__("re").findall(r"-?\d+\.?\d*(?:e[+-]?\d+)?", text.replace("tensor", ""))]
    rois = np.array(nums[: 4 * 12 * 4], dtype=np.float32).reshape(4, 12, 4)
    np.savez_compressed(path, rois=rois)
    print("wrote", path, rois.shape, "degenerate boxes:",
          int(((rois[..., 0] == rois[..., 2]) | (rois[..., 1] == rois[..., 3])).sum()))


if __name__ == "__main__":
    ref_stubs.install()
    torch.set_num_threads(8)
    gold = os.path.join(REPO, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    which = sys.argv[1:] or ["warp", "texture", "warp256", "warp_c1", "texture256", "roi", "cloth", "roiops", "modes", "channels", "depths", "depths_gp", "nonsquare", "pixel"]
    if "warp" in which:
        golden_warp(os.path.join(gold, "warp_step_64.npz"))
    if "texture" in which:
        golden_texture(os.path.join(gold, "texture_step_64.npz"))
    if "warp256" in which:          # BASELINE.json C2's resolution (bs 2: seconds on the CPU)
        golden_warp(os.path.join(gold, "warp_step_256.npz"), H=256, B=2, step_seeds=(100,), full="warp_256")
    if "warp_c1" in which:          # BASELINE.json C1: 64 x 64, bs 4
        golden_warp(os.path.join(gold, "warp_step_c1.npz"), H=64, B=4, step_seeds=(100, 101), full="warp_c1")
    if "texture256" in which:       # BASELINE.json C3's resolution: the depth-8 U-Net, 12 ROIs incl. a degenerate box
        golden_texture(os.path.join(gold, "texture_step_256.npz"), H=256, B=1, step_seeds=(200,), full="texture_256")
    if "roi" in which:
        golden_roi(os.path.join(gold, "notebook_rois.npz"))
    if "modes" in which:
        golden_warp_modes(os.path.join(gold, "warp_modes_64.npz"))
    if "channels" in which:
        golden_warp_channels(os.path.join(gold, "warp_channels_64.npz"))
    if "depths" in which:
        golden_warp_depths(os.path.join(gold, "warp_depths_64.npz"))
    if "depths_gp" in which:
        golden_warp_depths_gp(os.path.join(gold, "warp_depths_gp_64.npz"))
    if "pixel" in which:
        golden_warp_pixel(os.path.join(gold, "warp_pixel_64.npz"))
    if "nonsquare" in which:
        golden_warp_nonsquare(os.path.join(gold, "warp_nonsquare_128x64.npz"))
    if "roiops" in which:
        golden_roi_ops(os.path.join(gold, "roi_ops_reference.npz"))
    if "cloth" in which:
        golden_cloth_format(os.path.join(gold, "cloth_segment_reference.npz"),
                            os.path.join(gold, "cloth_segment_expected.npz"))
