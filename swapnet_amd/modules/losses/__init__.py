"""modules.losses of the reference: only PerceptualLoss is on the hot path
(/root/reference/modules/losses/perceptual.py); SSIM / Charbonnier are unused there."""
from .perceptual import PerceptualLoss  # noqa: F401
