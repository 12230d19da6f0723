"""PerceptualLoss (/root/reference/modules/losses/perceptual.py:13-79) = the VGG16 network of
the native texture model: content = sum over 5 slices of MSE between channel-L2-normalised
features, style = 5 x MSE of the image Gram matrices.  Inside a training step the arithmetic runs
fused in TextureModel.backward_G (texture.cpp); `criterion(output, target)` evaluates it standalone
(swn_model_perceptual).  This object is also the parameter view (net 2) through which VGG16 weights
are loaded (`load_vgg16_features(state_dict)` takes torchvision's
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
        """PerceptualLoss.forward (perceptual.py:49-66) -> (content, style) scalar tensors; style is 0 unless
        use_style.  backward() through either returns the library's image gradient (swn_model_perceptual)."""
        b, c, h, w = output.shape
        m = self._backend.ensure(b, h, w)
        content, style = _PerceptualFn.apply(output, target, m, bool(self.use_style))
        return content, (style if self.use_style else 0)

    __call__ = forward


import torch  # noqa: E402


class _PerceptualFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, output, target, model, use_style):
        out2, _ = model.perceptual(output, target, use_style)
        fctx.model, fctx.use_style = model, use_style
        fctx.save_for_backward(output.detach(), target.detach())
        fctx.src = (output.device, output.dtype)
        out2 = out2.to(output.device)
        return out2[0].clone(), out2[1].clone()

    @staticmethod
    def backward(fctx, g_content, g_style):
        output, target = fctx.saved_tensors
        _, d = fctx.model.perceptual(output, target, fctx.use_style, content_w=float(g_content),
                                     style_w=float(g_style) if g_style is not None else 0.0, want_grad=True)
        return d.to(device=fctx.src[0], dtype=fctx.src[1]), None, None, None
