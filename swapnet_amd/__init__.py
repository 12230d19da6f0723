"""swapnet_amd -- MI355X-native (gfx950) back end for the SwapNet two-stage GAN training hot
path.  The arithmetic lives in swapnet_amd/csrc (hand-written HIP behind a C ABI,
include/swapnet_hip.h); this package is the Python host side that mirrors the reference's
`models/`, `modules/` and `optimizers/` surface.

    import swapnet_amd
    swapnet_amd.install_as_reference_packages()   # then the reference's own train.py /
                                                   # inference.py / options/ run unchanged
"""
import sys

__version__ = "0.1.0"


def install_as_reference_packages():
    """Registers swapnet_amd.{models,modules,optimizers} under the reference's top-level package
    names, so `from models import create_model`, `models.get_options_modifier(...)`,
    `import optimizers` inside the reference's unchanged train.py / inference.py / options/
    resolve to the native implementation.  The reference's `datasets/`, `options/` and `util/`
    packages (dataloader, argparse, visualisation -- all out of the hot path) stay its own."""
    from . import models, modules, optimizers
    from .models import base_gan, base_model, texture_model, warp_model
    from .modules import discriminators, loss, losses, swapnet_modules
    from .util import decode_labels
    sys.modules["models"] = models
    sys.modules["models.base_model"] = base_model
    sys.modules["models.base_gan"] = base_gan
    sys.modules["models.warp_model"] = warp_model
    sys.modules["models.texture_model"] = texture_model
    sys.modules["modules"] = modules
    sys.modules["modules.swapnet_modules"] = swapnet_modules
    sys.modules["modules.discriminators"] = discriminators
    sys.modules["modules.loss"] = loss
    sys.modules["modules.losses"] = losses
    sys.modules["optimizers"] = optimizers
    sys.modules["util.decode_labels"] = decode_labels
