"""PatchGAN discriminator factory with the reference's signature
(/root/reference/modules/discriminators.py:45-136).  Inside a model step the native PatchGAN runs
fused on the zero-copy cat of condition and generator output; the module object is the parameter /
checkpoint view of it and `net(x)` evaluates it standalone through swn_model_discriminate."""
from .. import engine
from .native import NativeNet


class NLayerDiscriminator(NativeNet):
    """PatchGAN with `n_layers` stride-2 levels (3 = the 70x70 "basic" one), ndf=64, instance norm, bias on every conv
    (:91-136).  ndf is fixed at 64 as in the reference's only call site (models/base_gan.py:147)."""

    def __init__(self, backend, input_nc=22, ndf=64, n_layers=3):
        want = backend.cloth_channels + (backend.body_channels if backend.kind == "warp" else 3)
        if input_nc != want or ndf != 64:
            raise NotImplementedError("native PatchGAN: ndf 64, input channels = the stage's conditional "
                                      "input (%d here), got input_nc=%d ndf=%d" % (want, input_nc, ndf))
        if not 1 <= int(n_layers) <= 5:
            raise NotImplementedError("native PatchGAN: n_layers_D in [1, 5], got %d" % n_layers)
        if backend.models and backend.n_layers_D != int(n_layers):
            raise RuntimeError("the stage's networks already exist with n_layers_D = %d" % backend.n_layers_D)
        backend.n_layers_D = int(n_layers)
        super().__init__(backend, engine.NET_D)

    def forward(self, input):
        """NLayerDiscriminator.forward (:134-136): `input` = the conditioned batch in the reference's channel
        order, (B, 22, H, W) -> prediction map (B, 1, (H >> n_layers) - 2, (W >> n_layers) - 2).  Inference-only call on the current
        weights (inside a training step the discriminator runs fused in model.backward_D / backward_G)."""
        b, c, h, w = input.shape
        m = self._backend.ensure(b, h, w)
        return m.discriminate(input)

    __call__ = forward


class PixelDiscriminator(NativeNet):
    """1x1 PatchGAN ("pixelGAN", :139-175): Conv1x1(input_nc, 64) - LeakyReLU - Conv1x1(64, 128) - InstanceNorm - LeakyReLU -
    Conv1x1(128, 1); every conv carries a bias under instance norm (:152-155); state_dict keys net.{0,2,5}.{weight,bias}.  On the
    native side it is PatchGAN "depth 0" of the stage's context (swn_ctx_set_patchgan_layers(ctx, 0))."""

    def __init__(self, backend, input_nc=22, ndf=64):
        want = backend.cloth_channels + (backend.body_channels if backend.kind == "warp" else 3)
        if input_nc != want or ndf != 64:
            raise NotImplementedError("native PixelDiscriminator: ndf 64, input channels = the stage's conditional "
                                      "input (%d here), got input_nc=%d ndf=%d" % (want, input_nc, ndf))
        if backend.models and backend.n_layers_D != 0:
            raise RuntimeError("the stage's networks already exist with n_layers_D = %d" % backend.n_layers_D)
        backend.n_layers_D = 0
        super().__init__(backend, engine.NET_D)

    def forward(self, input):
        """PixelDiscriminator.forward (:172-174): (B, input_nc, H, W) -> (B, 1, H, W).  Inference-only call on the current weights."""
        b, c, h, w = input.shape
        return self._backend.ensure(b, h, w).discriminate(input)

    __call__ = forward


def define_D(input_nc, ndf, netD, n_layers_D=3, norm="batch", init_type="normal", init_gain=0.02, gpu_ids=[],
             backend=None):
    """discriminators.define_D (:45-88)."""
    if norm != "instance":
        raise NotImplementedError("normalization layer [%s] is not implemented (instance only)" % norm)
    if netD == "basic":
        return NLayerDiscriminator(backend, input_nc, ndf, n_layers=3)
    if netD == "n_layers":
        return NLayerDiscriminator(backend, input_nc, ndf, n_layers_D)
    if netD == "pixel":
        return PixelDiscriminator(backend, input_nc, ndf)
    raise NotImplementedError("Discriminator model name [%s] is not recognized" % netD)
