"""GANLoss of the reference (/root/reference/modules/loss.py:12-130).  The loss value and its
gradient are computed by the library (losses.hip, swn_op_gan_loss); this class reproduces what
is host-side in the reference: WHICH scalar target each call uses, drawn from the global torch
CPU RNG in the reference's order so seeded runs see identical labels.  `criterion(pred, is_real)`
is usable standalone like the reference's module: it returns a scalar tensor whose backward()
delivers the library's gradient (torch.autograd is glue only; no arithmetic of the loss runs in
torch)."""
import torch

from .. import engine


class _GanLossFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, pred, mode, label, is_real, lib_ctx):
        loss, dpred = engine.op_gan_loss(lib_ctx, pred, mode, label, is_real, 1.0, want_grad=pred.requires_grad)
        fctx.dpred, fctx.src = dpred, (pred.device, pred.dtype)
        return loss.reshape(()).to(pred.device)

    @staticmethod
    def backward(fctx, g):
        d = fctx.dpred
        if d is None:
            return None, None, None, None, None
        return (d * g.to(d.device)).to(device=fctx.src[0], dtype=fctx.src[1]), None, None, None, None


class GANLoss:
    default_real = 1.0
    default_fake = 0.0
    default_smooth_real = (0.7, 1.1)
    default_smooth_fake = (0.0, 0.3)
    MODES = {"vanilla": 0, "dragan": 0, "lsgan": 1, "wgan": 2, "wgan-gp": 2, "dragan-gp": 0, "dragan-lp": 0}
    GP_MODES = {"wgan-gp": 1, "dragan-gp": 2, "dragan-lp": 3}        # swn_hyper.gp_mode (modules/loss.py:133-184)

    def __init__(self, gan_mode, smooth_labels=True, target_real_label=None, target_fake_label=None):
        if gan_mode not in self.MODES:
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)      # e.g. mescheder-r1-gp, like loss.py:62
        self.gan_mode = gan_mode
        self.native_mode = self.MODES[gan_mode]
        self.gp_mode = self.GP_MODES.get(gan_mode, 0)
        self.smooth = smooth_labels
        self.real_label = target_real_label if target_real_label is not None else (
            self.default_smooth_real if smooth_labels else self.default_real)
        self.fake_label = target_fake_label if target_fake_label is not None else (
            self.default_smooth_fake if smooth_labels else self.default_fake)

    def to(self, device):
        return self

    def __call__(self, prediction, target_is_real):
        """GANLoss.__call__ (loss.py:110-130): draws the target scalar exactly like get_target_tensor, then
        BCE-with-logits / MSE / +-mean of `prediction` on the device.  Returns a scalar tensor."""
        label = self.sample_label(target_is_real)
        ctx = getattr(self, "ctx", None) or engine.default_context()
        return _GanLossFn.apply(prediction, self.native_mode, label, bool(target_is_real), ctx)

    forward = __call__

    @staticmethod
    def rand_between(low, high):
        """loss.py:65-77: torch.rand(1) * (high - low) + low in fp32 tensor arithmetic."""
        low, high = torch.tensor(low), torch.tensor(high)
        return float(torch.rand(1) * (high - low) + low)

    def sample_label(self, target_is_real):
        """get_target_tensor (loss.py:79-108).  With smooth labels BOTH branches sample from the
        REAL range -- the fake branch unpacks self.real_label (loss.py:102); reproduced as is.
        (The reference's hard mode crashes on len() of a 0-d tensor, loss.py:92; here it simply
        returns the constants.)"""
        if self.native_mode == 2:
            return 0.0
        label = self.real_label if target_is_real else self.fake_label
        if isinstance(label, (tuple, list)):
            low, high = self.real_label
            return self.rand_between(low, high)
        return float(label)
