"""Glue between the reference's nn.Module-shaped API and the library-owned networks.

A reference model holds `net_generator` / `net_discriminator` (nn.Modules) and two torch
optimizers.  Here all of them are views onto ONE `NativeBackend`, which owns the swn_model
handles (one per batch shape seen, all sharing the same flat parameter arenas' contents).
"""
import torch
from torch import nn

from .. import engine


class NativeBackend:
    """Owns the native model(s) of one GAN stage.  The arena layout of a network does not depend on the batch size, so when a
    differently-sized batch arrives (last batch of an epoch, batch_size=1 inference) a second swn_model is created ON THE FIRST
    ONE'S ARENAS (swn_model_create_shared): it adds its own activations only, and weights, gradients, Adam moments and step
    counters are simply the same memory -- nothing is copied when the batch size changes back and forth."""

    def __init__(self, kind, is_train=True, dropout=0.5, num_roi=12, ctx=None, lib=None, device=None,
                 default_shape=(1, 64, 64), body_channels=3, cloth_channels=19):
        self.kind, self.is_train, self.dropout, self.num_roi = kind, is_train, dropout, num_roi
        self.body_channels, self.cloth_channels = body_channels, cloth_channels
        self.default_shape = tuple(default_shape)
        self.n_layers_D = 3         # define_D(..., n_layers_D): set by NLayerDiscriminator before the first model exists
        self.ctx = ctx or engine.default_context(device=device, lib=lib)
        self.models = {}
        self.cur = None
        self.hyper = {}
        self.training = {engine.NET_G: True, engine.NET_D: True}
        self._pending = {}          # state dicts loaded before any shape is known

    # -- shape management ---------------------------------------------------------------
    def ensure(self, B, H, W):
        key = (int(B), int(H), int(W))
        m = self.models.get(key)
        if m is None:
            root = next(iter(self.models.values()), None)          # the first model owns the arenas, every later one shares them
            m = engine.NativeModel(self.ctx, self.kind, key[0], key[1], key[2], is_train=self.is_train,
                                   dropout=self.dropout, num_roi=self.num_roi, body_channels=self.body_channels,
                                   cloth_channels=self.cloth_channels, n_layers_D=self.n_layers_D, share=root)
            m.set_hyper(**self.hyper)
            self.models[key] = m
        if self.cur is None and self._pending:
            for (net, which), sd in list(self._pending.items()):
                m.load_state_dict(net, sd, which=which)
            self._pending = {}
        self.cur = m
        return m

    def _nets(self):
        return [engine.NET_G] + ([engine.NET_D] if self.is_train else []) + \
            ([engine.NET_VGG] if self.is_train and self.kind == "texture" else [])

    def any_model(self):
        """A model to answer shape-independent queries (parameter names / shapes)."""
        if self.cur is not None:
            return self.cur
        return self.ensure(*self.default_shape)

    def set_hyper(self, **kw):
        self.hyper.update(kw)
        for m in self.models.values():
            m.set_hyper(**self.hyper)

    # -- parameters ---------------------------------------------------------------------------
    def param_shapes(self, net):
        return self.any_model().param_infos(net)

    def state_dict(self, net, which=engine.W_WEIGHT):
        return self.any_model().state_dict(net, which=which, to_cpu=True)

    def load_state_dict(self, net, sd, which=engine.W_WEIGHT, strict=True):
        self.any_model().load_state_dict(net, sd, which=which, strict=strict)


class NativeNet(nn.Module):
    """nn.Module facade over one library-owned network (generator or discriminator).
    Keeps what the reference's BaseModel needs: state_dict() with the reference's keys,
    load_state_dict(), parameters() (for counting), train()/eval(), .cpu()/.cuda()/.to()."""

    def __init__(self, backend, net):
        super().__init__()
        object.__setattr__(self, "_backend", backend)
        self._net = net

    def native_param_shapes(self):
        return self._backend.param_shapes(self._net)

    def state_dict(self, *args, **kwargs):
        return self._backend.state_dict(self._net)

    def load_state_dict(self, state_dict, strict=True):
        self._backend.load_state_dict(self._net, state_dict, strict=strict)
        return self

    def parameters(self, recurse=True):
        for v in self._backend.state_dict(self._net).values():
            yield nn.Parameter(v, requires_grad=False)

    def named_parameters(self, prefix="", recurse=True):
        for k, v in self._backend.state_dict(self._net).items():
            yield prefix + k, nn.Parameter(v, requires_grad=False)

    def train(self, mode=True):
        self.training = mode
        self._backend.training[self._net] = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):        # the parameters live in the library's arenas on the GPU
        return self

    def cpu(self):
        return self

    def cuda(self, device=None):
        return self

    def extra_repr(self):
        shapes = self.native_param_shapes()
        return "native(%s): %d tensors, %.3f M parameters" % (
            self._backend.kind, len(shapes), sum(int(torch.tensor(s).prod()) for s in shapes.values()) / 1e6)
