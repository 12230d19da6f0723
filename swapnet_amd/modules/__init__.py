"""modules/__init__.py of the reference: weight init + norm-layer selection
(/root/reference/modules/__init__.py:7-74), re-targeted at library-owned parameters.

Initialisation stays host-side in torch: tensors are drawn with the
reference's own initialisers in the reference's order and uploaded through the C-ABI."""
import math
from collections import OrderedDict

import torch
from torch.nn import init


def init_tensor(t, init_type="normal", init_gain=0.02):
    """init_func of modules.init_weights applied to one weight tensor (modules/__init__.py:19-34)."""
    if init_type == "normal":
        init.normal_(t, 0.0, init_gain)
    elif init_type == "xavier":
        init.xavier_normal_(t, gain=init_gain)
    elif init_type == "kaiming":
        init.kaiming_normal_(t, a=0, mode="fan_in")
    elif init_type == "orthogonal":
        init.orthogonal_(t, gain=init_gain)
    else:
        raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
    return t


def _construction_order(layers):
    """Order in which the reference's layer constructors ran.  Plain modules are defined top to bottom
    (= state-dict order); the pix2pix U-Net is built recursively from the innermost block outwards, down
    conv before up conv in each block (modules/pix2pix_modules.py:113-177,208-246), while its state dict
    (module tree) lists the outermost block first."""
    unet = [n for n in layers if n.startswith("unet.")]
    if not unet:
        return list(layers)
    rest = [n for n in layers if not n.startswith("unet.")]           # `encode` is defined before the U-Net
    depth = lambda n: (len(n.split(".")) - 4) // 2                    # unet.model.model.K = block 0
    order = sorted(unet, key=lambda n: (-depth(n), int(n.rsplit(".", 1)[1])))
    return rest + order


def init_weights(net, init_type="normal", init_gain=0.02):
    """modules.init_weights (modules/__init__.py:7-45) for a swapnet_amd network: `init_type` on every
    Conv*/ConvTranspose* weight, bias := 0.  `net` is any module exposing `native_param_shapes()` /
    `load_state_dict()`.

    The global CPU RNG is consumed exactly like the reference does when it builds and initialises the
    same network -- every torch layer constructor draws its default weight (kaiming_uniform) and bias
    (uniform) in construction order, then init_weights re-draws the weights in module-tree order -- so
    `torch.manual_seed(s); create_model(opt)` starts from the reference's weights, bit for bit
    (tests/test_models_api.py::test_seeded_init_equals_the_reference)."""
    print("initialize network with %s" % init_type)
    shapes = net.native_param_shapes()
    layers = OrderedDict()
    for name, shape in shapes.items():
        prefix, kind = name.rsplit(".", 1)
        layers.setdefault(prefix, {})[kind] = shape
    for prefix in _construction_order(layers):                        # nn.Conv2d / nn.ConvTranspose2d.reset_parameters
        init.kaiming_uniform_(torch.empty(layers[prefix]["weight"]), a=math.sqrt(5))
        if "bias" in layers[prefix]:
            torch.empty(layers[prefix]["bias"]).uniform_(-1.0, 1.0)
    sd = {}
    for name, shape in shapes.items():
        if name.endswith(".bias"):
            sd[name] = torch.zeros(shape)
        else:
            sd[name] = init_tensor(torch.empty(shape), init_type, init_gain)
    net.load_state_dict(sd)


class Identity(torch.nn.Module):
    def forward(self, x):
        return x


def get_norm_layer(norm_type="instance"):
    """Only instance norm (the reference's default, base_gan.py:72-77) is implemented natively."""
    if norm_type == "instance":
        return "instance"
    if norm_type in ("batch", "none"):
        raise NotImplementedError("normalization layer [%s] is not implemented in swapnet_amd "
                                  "(default --norm instance only)" % norm_type)
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)
