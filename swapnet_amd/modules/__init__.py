"""modules/__init__.py of the reference: weight init + norm-layer selection
(/root/reference/modules/__init__.py:7-74), re-targeted at library-owned parameters.

Initialisation stays host-side in torch: tensors are drawn with the
reference's own initialisers in the reference's order and uploaded through the C-ABI."""
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


def init_weights(net, init_type="normal", init_gain=0.02):
    """Initialise every Conv*/ConvTranspose* weight of a swapnet_amd network, bias := 0.
    `net` is any module exposing `native_param_shapes()` / `load_state_dict()`."""
    print("initialize network with %s" % init_type)
    sd = {}
    for name, shape in net.native_param_shapes().items():
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
