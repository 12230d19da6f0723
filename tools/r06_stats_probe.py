"""Forward accuracy of the >1024-pixel InstanceNorm levels against the float64 oracle, with the conv-epilogue statistics on / off
(SWN_CONV_STATS, read when a model is built).  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import swapnet_oracle as O
from swapnet_amd import engine

def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / b.double().norm())

ctx = engine.Context(workspace_mb=1024)
B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 256
torch.manual_seed(3)
G = O.warp_module_params()
batch = O.synth_warp_batch(B, H, H, seed=99)
taps64 = {}
with torch.no_grad():
    O.warp_module_forward({k: v.double() for k, v in G.items()}, batch[0].double(), batch[1].double(), taps=taps64)
for v in ("1", "0"):
    os.environ["SWN_CONV_STATS"] = v
    m = engine.NativeModel(ctx, "warp", B, H, H, is_train=False)
    m.load_state_dict(engine.NET_G, G)
    m.set_input(0, batch[0]); m.set_input(1, batch[1])
    m.forward(False, 0)
    print("SWN_CONV_STATS=" + v, {k: "%.2e" % rel(m.tap(engine.NET_G, k)[:, :taps64[k].shape[1]], taps64[k]) for k in ("body_d1", "body_d2", "cloth_d2", "body_d3", "cloth_d3", "res0", "dual_u2")})
    m.close()
