"""Diagnostic (GPU): per-tensor gradient error of the native warp step against the float64 oracle, next to the error
of torch's own fp32 CPU backward, under the kernel-selection switches.  Usage: python tools/grad_error_table.py [H] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import swapnet_oracle as O          # noqa: E402
from swapnet_amd import engine                  # noqa: E402
from tests import backends                      # noqa: E402
from tests.test_warp_step import noise_bias     # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
labels = [0.9, 0.8, 1.0]
torch.manual_seed(3)
G, D = O.warp_module_params(), O.patchgan_params(22)
batch = O.synth_warp_batch(B, H, H, seed=99)
s32 = O.WarpStepOracle(G, D); s32.step(*batch, labels=labels)
s64 = O.WarpStepOracle(G, D, dtype=torch.float64); s64.step(*batch, labels=labels)
rows = {}
for tag, env in (("default", {}), ("wino_k4=0", {"SWN_WINO_K4": "0"}), ("winograd=0", {"SWN_WINOGRAD": "0"})):
    for k in ("SWN_WINO_K4", "SWN_WINOGRAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = engine.Context(workspace_mb=1024)
    m = engine.NativeModel(ctx, "warp", B, H, H)
    backends.reset_state(m, {0: G, 1: D})
    for i, t in enumerate(batch):
        m.set_input(i, t)
    m.forward(False, 0); m.backward_D(labels[0], labels[1])
    gD = m.state_dict(1, which=engine.W_GRAD, to_cpu=True)
    m.optimizer_step(1); m.backward_G(labels[2])
    gG = m.state_dict(0, which=engine.W_GRAD, to_cpu=True)
    for which, got, r64, r32 in (("D", gD, s64.grads_D, s32.grads_D), ("G", gG, s64.grads_G, s32.grads_G)):
        for k, v in r64.items():
            if noise_bias(k):
                continue
            rows.setdefault(which + ":" + k, {"torch32": backends.rel_l2(r32[k], v)})[tag] = backends.rel_l2(got[k], v)
    m.close(); ctx.close()
print("%-48s %9s %9s %9s %9s" % ("tensor", "torch32", "default", "wino_k4=0", "winograd=0"))
for k, r in rows.items():
    print("%-48s %9.2e %9.2e %9.2e %9.2e" % (k, r["torch32"], r["default"], r["wino_k4=0"], r["winograd=0"]))
