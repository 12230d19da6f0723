"""SURVEY.md 8(b) "Module signatures": the reference's loss / discriminator modules are callable standalone,
each as ONE native call through the C-ABI (no torch arithmetic), and agree with the oracle:
  GANLoss(gan_mode, smooth_labels)(pred, target_is_real)           modules/loss.py:110-130
  PerceptualLoss(use_style)(output, target) -> (content, style)    modules/losses/perceptual.py:49-66
  NLayerDiscriminator.forward(input)                               modules/discriminators.py:134-136
torch.autograd only carries the library's gradient back to the caller (north_star: "autograd glue")."""
import pytest
import torch

from oracle import swapnet_oracle as O
from swapnet_amd import engine
from swapnet_amd.modules.loss import GANLoss
from tests import backends
from tests.test_models_api import make_opt

pytestmark = pytest.mark.small_channel_winograd      # tests/conftest.py: small shapes on the Winograd forms

BACKENDS = [pytest.param("sim", id="hostsim"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _ctx(kind):
    return backends.gpu_ctx() if kind == "gpu" else backends.hostsim_ctx()


def rel(a, b):
    return backends.rel_l2(a, b)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", ["vanilla", "lsgan", "wgan"])
@pytest.mark.parametrize("is_real", [True, False])
def test_ganloss_call(backend, mode, is_real):
    ctx = _ctx(backend)
    crit = GANLoss(mode, smooth_labels=True)
    crit.ctx = ctx
    g = torch.Generator().manual_seed(3)
    pred = torch.randn(4, 1, 6, 6, generator=g, requires_grad=True)
    torch.manual_seed(11)
    loss = crit(pred, is_real)
    loss.backward()
    # the reference: same RNG draw (both branches sample the REAL range, loss.py:93,102), same reduction
    torch.manual_seed(11)
    p2 = pred.detach().clone().requires_grad_(True)
    label = O.smooth_label() if mode != "wgan" else torch.zeros(1)
    ref = O.gan_loss(p2, label, mode, is_real)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref)) + 1e-7
    assert rel(pred.grad, p2.grad) < 1e-5
    # the RNG was advanced exactly like the reference's (one draw, none for wgan)
    a = torch.rand(1)
    torch.manual_seed(11)
    if mode != "wgan":
        O.smooth_label()
    assert torch.equal(a, torch.rand(1))


def test_ganloss_rejects_unknown_modes_like_the_reference():
    with pytest.raises(NotImplementedError):
        GANLoss("hinge")


@pytest.mark.parametrize("backend", BACKENDS)
def test_discriminator_forward_call(backend, tmp_path):
    from swapnet_amd.models import create_model
    opt = make_opt(tmp_path, backend)
    model = create_model(opt)
    torch.manual_seed(5)
    D = O.patchgan_params(22)
    model.net_discriminator.load_state_dict(D)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 22, 64, 64, generator=g)
    bodys, inputs, targets = O.synth_warp_batch(2, 64, 64, seed=1234)
    model.eval()
    model.set_input(dict(bodys=bodys, input_cloths=inputs, target_cloths=targets, cloth_paths=["", ""], body_paths=["", ""]))
    model.forward()
    before = model.fakes.clone()
    pred = model.net_discriminator(x)
    with torch.no_grad():
        ref = O.patchgan_forward(D, x)
    assert tuple(pred.shape) == tuple(ref.shape) == (2, 1, 6, 6)
    assert rel(pred, ref) < 1e-3
    # the standalone call leaves the model's own buffers alone
    model._fakes = None
    assert torch.equal(model.fakes, before)
    # and sees weight updates: after a training step the same input gives the updated D's prediction
    torch.manual_seed(1)
    model.optimize_parameters()
    pred2 = model.net_discriminator(x)
    with torch.no_grad():
        ref2 = O.patchgan_forward({k: v.cpu() for k, v in model.net_discriminator.state_dict().items()}, x)
    assert rel(pred2, ref2) < 1e-3 and rel(pred2, pred) > 1e-4


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("use_style", [True, False])
def test_perceptual_loss_call(backend, use_style, tmp_path):
    from swapnet_amd.models import create_model
    opt = make_opt(tmp_path, backend, model="texture", lambda_style=1e-8 if use_style else 0.0)
    with pytest.warns(RuntimeWarning, match="SEEDED RANDOM VGG16"):
        model = create_model(opt)
    assert model.vgg_source == "seeded-random"
    crit = model.criterion_perceptual
    vgg = O.vgg16_feature_params()
    names = list(crit.native_param_shapes().keys())
    crit.load_state_dict({names[2 * i + j]: t for i, wb in enumerate(vgg) for j, t in enumerate(wb)})
    g = torch.Generator().manual_seed(2)
    out = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).requires_grad_(True)
    tgt = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    content, style = crit(out, tgt)
    o2 = out.detach().clone().requires_grad_(True)
    rc, rs = O.perceptual_loss(vgg, o2, tgt, use_style=use_style)
    assert abs(float(content) - float(rc)) <= 1e-3 * abs(float(rc))
    if use_style:
        assert abs(float(style) - float(rs)) <= 1e-3 * abs(float(rs))
        (20.0 * content + 1e-8 * style).backward()
        (20.0 * rc + 1e-8 * rs).backward()
    else:
        assert style == 0
        (20.0 * content).backward()
        (20.0 * rc).backward()
    assert rel(out.grad, o2.grad) < 2e-3


def test_vgg_weights_file_is_loaded(tmp_path):
    """a --vgg_weights file (torchvision vgg16 layout) replaces the seeded-random stand-in, silently."""
    import warnings
    from swapnet_amd.models import create_model
    vgg = O.vgg16_feature_params(seed=77)
    idx = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
    sd = {}
    for i, (w, b) in zip(idx, vgg):
        sd["features.%d.weight" % i], sd["features.%d.bias" % i] = w, b
    path = str(tmp_path / "vgg16.pth")
    torch.save(sd, path)
    opt = make_opt(tmp_path, "sim", model="texture", vgg_weights=path)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model = create_model(opt)
    assert model.vgg_source == path
    got = model.criterion_perceptual.state_dict()
    assert torch.equal(got["net.0.0.weight"].cpu(), vgg[0][0]) and torch.equal(got["net.4.28.bias"].cpu(), vgg[12][1])


@pytest.mark.gpu
def test_style_term_with_a_large_batch_uses_the_generic_gram_kernels(tmp_path):
    """batch_size * 3 > 128 rows of the image Gram (bs >= 43) used to raise; the tiled Gram kernels have no
    such limit.  PerceptualLoss(use_style=True) at bs 44 against the oracle, value and gradient."""
    from swapnet_amd.models import create_model
    opt = make_opt(tmp_path, "gpu", model="texture", batch_size=44)
    with pytest.warns(RuntimeWarning):
        model = create_model(opt)
    crit = model.criterion_perceptual
    vgg = O.vgg16_feature_params()
    names = list(crit.native_param_shapes().keys())
    crit.load_state_dict({names[2 * i + j]: t for i, wb in enumerate(vgg) for j, t in enumerate(wb)})
    g = torch.Generator().manual_seed(6)
    out = (torch.rand(44, 3, 64, 64, generator=g) * 2 - 1).requires_grad_(True)
    tgt = torch.rand(44, 3, 64, 64, generator=g) * 2 - 1
    content, style = crit(out, tgt)
    o2 = out.detach().clone().requires_grad_(True)
    rc, rs = O.perceptual_loss(vgg, o2, tgt, use_style=True)
    assert abs(float(style) - float(rs)) <= 1e-3 * abs(float(rs)) and abs(float(content) - float(rc)) <= 1e-3 * abs(float(rc))
    (1e-8 * style).backward()
    (1e-8 * rs).backward()
    assert rel(out.grad, o2.grad) < 2e-3
