"""PerceptualLoss (/root/reference/modules/losses/perceptual.py:13-79) = the VGG16 network of
the native texture model: content = sum over 5 slices of MSE between channel-L2-normalised
features, style = 5 x MSE of the image Gram matrices.  The arithmetic runs inside
TextureModel.backward_G (texture.cpp); this object is the parameter view (net 2) through which
VGG16 weights are loaded (`load_vgg16_features(state_dict)` takes torchvision's
vgg16().features.state_dict() keys: "<idx>.weight" / "<idx>.bias")."""
from .. import native
from ... import engine

VGG16_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
SLICES = [(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)]            # perceptual.py:28-34


class PerceptualLoss(native.NativeNet):
    def __init__(self, normalize=True, use_style=False, backend=None):
        if not normalize:
            raise NotImplementedError("PerceptualLoss(normalize=False) is not implemented natively")
        super().__init__(backend, engine.NET_VGG)
        self.normalize, self.use_style = normalize, use_style

    @staticmethod
    def _slice_of(idx):
        for s, (lo, hi) in enumerate(SLICES):
            if lo <= idx < hi:
                return s
        raise KeyError(idx)

    def load_vgg16_features(self, features_state_dict):
        sd = {}
        for idx in VGG16_CONV_IDX:
            for kind in ("weight", "bias"):
                sd["net.%d.%d.%s" % (self._slice_of(idx), idx, kind)] = features_state_dict["%d.%s" % (idx, kind)]
        self.load_state_dict(sd)

    def forward(self, output, target):
        raise NotImplementedError("content/style terms are evaluated inside TextureModel.backward_G "
                                  "(loss_G_content / loss_G_style)")
