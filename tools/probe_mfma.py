import ctypes, sys
sys.path.insert(0, '.')
from swapnet_amd import engine
ctx = engine.Context(device=0, workspace_mb=64)
for it in (1024, 4096, 16384):
    for z in (0, 1):
        o = (ctypes.c_float * 4)()
        ctx.lib.call("swn_probe_mfma", ctx.handle, z, it, o)
        print("iters", it, "zeros", z, "fp16 TFLOP/s %.1f  clock %.3f GHz  ms %.3f  pipe %.3f" % tuple(o))
