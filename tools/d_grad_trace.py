"""Diagnostic (GPU): where along the discriminator's backward pass does the native gradient drift from the float64
oracle faster than torch's fp32 CPU backward does?  Prints rel-L2 error of d(loss_D)/d(activation) per layer."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import swapnet_oracle as O          # noqa: E402
from swapnet_amd import engine                  # noqa: E402
from tests import backends                      # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
labels = [0.9, 0.8, 1.0]
torch.manual_seed(3)
G, D = O.warp_module_params(), O.patchgan_params(22)
batch = O.synth_warp_batch(B, H, H, seed=99)


def oracle_taps(dtype):
    Gd = {k: v.to(dtype) for k, v in G.items()}
    Dd = {k: v.to(dtype).requires_grad_(True) for k, v in D.items()}
    bodys, inputs, targets = [t.to(dtype) for t in batch]
    with torch.no_grad():
        fakes = O.warp_module_forward(Gd, bodys, inputs)
    x = torch.cat((torch.cat((bodys, fakes), 1), torch.cat((bodys, targets), 1)), 0)       # [fake | real] like the 2B batch
    taps = {}
    pred = O.patchgan_forward(Dd, x, taps=taps)
    taps["pred"] = pred
    for t in taps.values():
        t.retain_grad()
    lf = O.gan_loss(pred[:B], torch.tensor([labels[0]], dtype=dtype))
    lr = O.gan_loss(pred[B:], torch.tensor([labels[1]], dtype=dtype))
    (0.5 * (lf + lr)).backward()
    return {k: (v.detach(), v.grad.detach()) for k, v in taps.items()}, {k: v.grad for k, v in Dd.items()}


t64, g64 = oracle_taps(torch.float64)
t32, g32 = oracle_taps(torch.float32)
ctx = engine.Context(workspace_mb=1024)
m = engine.NativeModel(ctx, "warp", B, H, H)
backends.reset_state(m, {0: G, 1: D})
for i, t in enumerate(batch):
    m.set_input(i, t)
m.forward(False, 0)
m.backward_D(labels[0], labels[1])
gD = m.state_dict(1, which=engine.W_GRAD, to_cpu=True)
print("%-8s %-26s %10s %10s | %10s %10s" % ("tap", "shape", "act t32", "act hip", "grad t32", "grad hip"))
for k in ("pred", "d3", "d2", "d1", "d0"):
    a64, d64 = t64[k]
    a32, d32 = t32[k]
    ah = m.tap(engine.NET_D, k).cpu()[:, :a64.shape[1]]
    dh = m.tap_grad(engine.NET_D, k).cpu()[:, :a64.shape[1]]
    print("%-8s %-26s %10.2e %10.2e | %10.2e %10.2e" % (k, tuple(a64.shape), backends.rel_l2(a32, a64), backends.rel_l2(ah, a64),
                                                         backends.rel_l2(d32, d64), backends.rel_l2(dh, d64)))
for k in g64:
    print("%-24s weight-grad  t32 %.2e  hip %.2e" % (k, backends.rel_l2(g32[k], g64[k]), backends.rel_l2(gD[k], g64[k])))
