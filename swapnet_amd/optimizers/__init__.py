"""optimizers of the reference (/root/reference/optimizers/__init__.py): option modifiers and
`define_optimizer`, returning an object with the torch.optim.Optimizer surface the reference
uses (zero_grad / step / state_dict / load_state_dict / param_groups) backed by the fused HIP
AdamW over the network's flat arena.  AdaBound (non-default; its package is not even
installable offline) is not implemented."""
from argparse import ArgumentParser
from collections import OrderedDict

from .. import engine


def get_options_modifier(optimizer_name):
    optimizer_name = optimizer_name.lower()
    if "adam" in optimizer_name:
        return adam_modifier
    if "adabound" in optimizer_name:
        return adabound_modifier
    raise NotImplementedError


def adam_modifier(parser: ArgumentParser, *_):
    parser.add_argument("--b1", type=float, default=0.9, help="Adam b1")
    parser.add_argument("--b2", type=float, default=0.999, help="Adam b2")
    return parser


def adabound_modifier(parser: ArgumentParser, *_):
    parser = adam_modifier(parser)
    parser.add_argument("--final_lr", type=float, default=0.1, help="AdaBound final_lr")
    return parser


class NativeAdamW:
    """torch.optim.AdamW(lr, betas, eps=1e-8, weight_decay, amsgrad=False) over one arena."""

    def __init__(self, backend, net, lr, weight_decay, betas):
        self.backend, self.net = backend, net
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay, betas=tuple(betas), eps=1e-8, amsgrad=False)]
        self._pushed = None
        self.sync()

    def sync(self):
        """param_groups is live like torch's: a scheduler (or the user) editing param_groups[0]["lr"] /
        ["weight_decay"] / ["betas"] takes effect at the next step.  Each optimizer owns its own hyper-parameters
        (G: lr, weight_decay, b1, b2; D: d_lr, d_weight_decay, d_b1, d_b2), so loading optimizer_D's state never
        touches G's betas."""
        g = self.param_groups[0]
        cur = (float(g["lr"]), float(g["weight_decay"]), float(g["betas"][0]), float(g["betas"][1]))
        if cur != self._pushed:
            key = "d_" if self.net == engine.NET_D else ""
            self.backend.set_hyper(**{key + "lr": cur[0], key + "weight_decay": cur[1], key + "b1": cur[2], key + "b2": cur[3]})
            self._pushed = cur

    def zero_grad(self, set_to_none=False):
        # every backward pass overwrites the whole gradient arena (each parameter has exactly one
        # producer and the planner's first write is a plain store), so there is nothing to clear
        pass

    def step(self, closure=None):
        self.sync()
        self.backend.cur.optimizer_step(self.net)

    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout: state[i] = {step, exp_avg, exp_avg_sq}."""
        m = self.backend.any_model()
        names = list(m.param_infos(self.net).keys())
        avg = m.state_dict(self.net, which=engine.W_EXP_AVG, to_cpu=True)
        sq = m.state_dict(self.net, which=engine.W_EXP_AVG_SQ, to_cpu=True)
        step = m.optim_step_count(self.net)
        import torch
        state = OrderedDict()
        if step > 0:
            for i, n in enumerate(names):
                state[i] = dict(step=torch.tensor(float(step)), exp_avg=avg[n], exp_avg_sq=sq[n])
        groups = [dict(self.param_groups[0], params=list(range(len(names))))]
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        m = self.backend.any_model()
        names = list(m.param_infos(self.net).keys())
        state = sd["state"]
        if state:
            avg = {n: state[i]["exp_avg"] for i, n in enumerate(names)}
            sq = {n: state[i]["exp_avg_sq"] for i, n in enumerate(names)}
            m.load_state_dict(self.net, avg, which=engine.W_EXP_AVG)
            m.load_state_dict(self.net, sq, which=engine.W_EXP_AVG_SQ)
            m.optim_step_count(self.net, int(float(state[0]["step"])))
        g = sd["param_groups"][0]
        self.param_groups[0].update(lr=g["lr"], weight_decay=g["weight_decay"], betas=tuple(g["betas"]))
        self.sync()


def define_optimizer(parameters, opt, net: str):
    """optimizers.define_optimizer (:37-60).  `parameters` is the NativeNet (or its
    .parameters() generator is ignored): the optimizer binds to the net's backend."""
    if net != "D" and net != "G":
        raise ValueError(f"net arg must be 'D' or 'G', received {net}")
    choice = getattr(opt, "optimizer_" + net)
    if choice != "AdamW":
        raise NotImplementedError("optimizer %s is not implemented natively (AdamW only)" % choice)
    lr = opt.d_lr if net == "D" else opt.lr
    wd = opt.d_weight_decay if net == "D" else opt.weight_decay
    backend = getattr(parameters, "_backend", None)
    if backend is None:
        raise ValueError("define_optimizer expects the swapnet_amd network object (not .parameters())")
    return NativeAdamW(backend, engine.NET_D if net == "D" else engine.NET_G, lr, wd, (opt.b1, opt.b2))
