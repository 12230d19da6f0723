"""PatchGAN discriminator factory with the reference's signature
(/root/reference/modules/discriminators.py:45-136).  The native PatchGAN always runs
conditioned inside a model step (its input buffer is the zero-copy cat of condition and
generator output), so the module object is the parameter / checkpoint view of it."""
from .. import engine
from .native import NativeNet


class NLayerDiscriminator(NativeNet):
    """70x70 PatchGAN, n_layers=3, ndf=64, instance norm, bias on every conv (:91-136)."""

    def __init__(self, backend, input_nc=22, ndf=64, n_layers=3):
        if input_nc != 22 or ndf != 64 or n_layers != 3:
            raise NotImplementedError("native PatchGAN is the reference default: 22 input channels, ndf 64, 3 layers")
        super().__init__(backend, engine.NET_D)

    def forward(self, input):
        raise NotImplementedError(
            "the native PatchGAN runs inside model.backward_D / backward_G on the conditioned buffer; "
            "its prediction map is available as model.backend.cur.tap(NET_D, 'pred')")


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
        raise NotImplementedError("Discriminator model name [pixel] is not implemented natively")
    raise NotImplementedError("Discriminator model name [%s] is not recognized" % netD)
