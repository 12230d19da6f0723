"""WarpModule / TextureModule with the reference's constructor + forward signatures
(/root/reference/modules/swapnet_modules.py:22-260), executed by the HIP library.

Standalone use (inference, parity tests):
    net = WarpModule(body_channels=3, cloth_channels=19)
    net.load_state_dict(torch.load("latest_net_generator.pth"))     # reference checkpoint keys
    fakes = net(body, cloth)                                        # (B,19,H,W) in (-1,1)
Inside WarpModel / TextureModel the same object is a view onto the model's NativeBackend.
"""
import math

import torch

from .. import engine
from .native import NativeBackend, NativeNet


class WarpModule(NativeNet):
    """Dual-encoder U-Net with 4 residual blocks (swapnet_modules.py:28-151)."""

    def __init__(self, body_channels=3, cloth_channels=19, dropout=0.5, backend=None, lib=None):
        if backend is not None and (backend.body_channels, backend.cloth_channels) != (body_channels, cloth_channels):
            raise ValueError("WarpModule(%d, %d) on a backend built for (%d, %d) channels" % (
                body_channels, cloth_channels, backend.body_channels, backend.cloth_channels))
        backend = backend or NativeBackend("warp", is_train=False, dropout=dropout, lib=lib,
                                           body_channels=body_channels, cloth_channels=cloth_channels)
        super().__init__(backend, engine.NET_G)
        self.training = True

    def forward(self, body, cloth):
        B, _, H, W = body.shape
        m = self._backend.ensure(B, H, W)
        m.set_input(0, body)
        m.set_input(1, cloth)
        m.forward(training=self.training, seed=int(torch.randint(0, 2 ** 31 - 1, (1,))) if self.training else 0)
        return m.output()


class TextureModule(NativeNet):
    """RoIAlign -> UNetDown(36,36) -> nearest upsample -> cat cloth -> pix2pix U-Net
    (swapnet_modules.py:154-260).  Only the default configuration is native: instance norm,
    unet_type="pix2pix", depth log2(img_size)."""

    def __init__(self, texture_channels=3, cloth_channels=19, num_roi=12, norm_type="batch", dropout=0.5,
                 unet_type="pix2pix", img_size=128, backend=None, lib=None):
        if norm_type != "instance":
            raise NotImplementedError("normalization layer [%s] is not implemented (instance only)" % norm_type)
        if unet_type != "pix2pix":
            raise NotImplementedError("unet_type [%s] is not implemented" % unet_type)
        if texture_channels != 3:
            raise ValueError("swapnet_amd TextureModule supports 3 texture channels (the VGG16 perceptual loss and the "
                             "image Gram are defined on RGB)")
        if backend is not None and backend.cloth_channels != cloth_channels:
            raise ValueError("TextureModule(cloth_channels=%d) on a backend built for %d" % (cloth_channels, backend.cloth_channels))
        self.img_size = img_size
        self.num_downs = math.frexp(img_size)[1] - 1          # swapnet_modules.py:178
        backend = backend or NativeBackend("texture", is_train=False, dropout=dropout, num_roi=num_roi, lib=lib,
                                           default_shape=(1, img_size, img_size), cloth_channels=cloth_channels)
        super().__init__(backend, engine.NET_G)
        self.num_roi = num_roi
        self.training = True

    @staticmethod
    def reshape_rois(rois):
        """(B,R,4) -> (B*R,5) with the batch index in column 0 (swapnet_modules.py:210-229)."""
        B, R = rois.shape[0], rois.shape[1]
        b_idx = torch.arange(B).unsqueeze(-1).expand(B, R).reshape(-1, 1).to(rois.device).type(rois.dtype)
        return torch.cat((b_idx, rois.reshape(-1, rois.shape[-1])), dim=1)

    def forward(self, input_tex, rois, cloth):
        B, _, H, W = input_tex.shape
        if H != self.img_size:
            raise ValueError("TextureModule was built for img_size=%d, got %d" % (self.img_size, H))
        m = self._backend.ensure(B, H, W)
        m.set_input(0, input_tex)
        m.set_input(1, rois)
        m.set_input(2, cloth)
        m.forward(training=self.training, seed=int(torch.randint(0, 2 ** 31 - 1, (1,))) if self.training else 0)
        return m.output()
