"""Diagnostic (GPU): where the host->device time of `set_input` goes.  Times, for the C2 batch (bodys 25 MB + two one-hot
cloth tensors of 159 MB), (a) bare torch copies pageable / pinned on the current and on a side stream, (b) the
library's upload + layout kernel alone, (c) uploads interleaved with training steps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import swapnet_oracle as O          # noqa: E402
from swapnet_amd import engine                  # noqa: E402

B, H = 32, 256
batch = O.synth_warp_batch(B, H, H, seed=1)
host = [t.contiguous() for t in batch]
pinned = [t.pin_memory() for t in host]
nbytes = sum(t.numel() * 4 for t in host)
dev = [torch.empty_like(t, device="cuda") for t in host]
side = torch.cuda.Stream()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def copies(src, stream=None, nb=True):
    def f():
        if stream is None:
            for d, s in zip(dev, src):
                d.copy_(s, non_blocking=nb)
        else:
            with torch.cuda.stream(stream):
                for d, s in zip(dev, src):
                    d.copy_(s, non_blocking=nb)
    return f


for name, src in (("pageable", host), ("pinned", pinned)):
    for sname, st in (("current stream", None), ("side stream", side)):
        dt = timeit(copies(src, st))
        print("bare copies %-9s %-15s %7.2f ms  %6.1f GB/s" % (name, sname, dt * 1e3, nbytes / dt / 1e9))

ctx = engine.Context(workspace_mb=1024)
m = engine.NativeModel(ctx, "warp", B, H, H)
m.set_hyper()
for name, src in (("pageable", host), ("pinned", pinned)):
    dt = timeit(lambda: [m.set_input(i, t) for i, t in enumerate(src)])
    print("set_input x3 %-9s %7.2f ms  %6.1f GB/s" % (name, dt * 1e3, nbytes / dt / 1e9))
labels = [0.9, 0.8, 1.0]
dt0 = timeit(lambda: m.step(labels, training=True, seed=1))
print("step alone %.2f ms" % (dt0 * 1e3))
for name, src in (("pageable", host), ("pinned", pinned)):
    dt = timeit(lambda: ([m.set_input(i, t) for i, t in enumerate(src)], m.step(labels, training=True, seed=1)))
    print("set_input x3 + step %-9s %7.2f ms (step alone %.2f)" % (name, dt * 1e3, dt0 * 1e3))
# host-side time of the enqueue only (is the caller blocked?)
for name, src in (("pageable", host), ("pinned", pinned)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, t in enumerate(src):
        m.set_input(i, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("set_input x3 %-9s host enqueue %.2f ms, drained after %.2f ms" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
