"""Diagnostic (GPU): accumulation error of the conv kernels against float64 -- the library's kernel, its naive
one-thread-per-output fmaf chain (a sequential round-to-nearest reference on the same device) and torch's fp32 CPU
convolution, on PatchGAN's shapes.  Usage: python tools/acc_error.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swapnet_amd import engine                  # noqa: E402
from tests import backends                      # noqa: E402
from tests.test_ops import run_conv, ref_conv, K4S2, K4S1, K3REFL     # noqa: E402

ctx = engine.Context(workspace_mb=1024)
g = torch.Generator().manual_seed(0)
print("%-34s %8s %10s %10s %10s" % ("shape", "K", "torch32", "naive", "kernel"))
for kind, n, ci, h, co in ((K4S2, 4, 64, 128, 128), (K4S2, 4, 128, 64, 256), (K4S1, 4, 256, 32, 512), (K4S2, 2, 512, 16, 512),
                           (K3REFL, 2, 256, 64, 256), (K3REFL, 2, 1024, 16, 1024)):
    k = 3 if kind == K3REFL else 4
    x = torch.randn(n, ci, h, h, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    r64 = ref_conv(x.double(), w.double(), None, kind, 0)
    r32 = ref_conv(x, w, None, kind, 0)
    nv = run_conv(ctx, kind, 0, 0, True, x, w, None, 0, r32.shape)
    kr = run_conv(ctx, kind, 0, 0, False, x, w, None, 0, r32.shape)
    print("%-34s %8d %10.2e %10.2e %10.2e" % ("k%d n%d ci%d h%d co%d" % (k, n, ci, h, co), ci * k * k, backends.rel_l2(r32, r64),
                                               backends.rel_l2(nv, r64), backends.rel_l2(kr, r64)))
