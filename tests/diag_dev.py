import sys, torch
sys.path.insert(0, ".")
from swapnet_amd import engine, _C
from oracle import swapnet_oracle as O
from tests.test_ops import run_conv, ref_conv, rel
ctx = engine.Context(workspace_mb=1024)
B, H = 2, 64
torch.manual_seed(0)
G = O.warp_module_params()
batch = O.synth_warp_batch(B, H, H, seed=1234)
taps = {}
with torch.no_grad(): O.warp_module_forward(G, batch[0], batch[1], taps=taps)
m = engine.NativeModel(ctx, "warp", B, H, H, is_train=False)
m.load_state_dict(0, G); m.set_input(0, batch[0]); m.set_input(1, batch[1]); m.forward(False, 0)
for name, ref in taps.items():
    t = m.tap(0, name).cpu()
    for c0 in range(0, ref.shape[1], max(ref.shape[1]//3, 1)):
        c1 = min(c0 + max(ref.shape[1]//3,1), ref.shape[1])
        print(name, c0, c1, rel(t[:, c0:c1], ref[:, c0:c1]))
g = torch.Generator().manual_seed(0)
for (n, ci, h, co) in [(2,1024,4,256),(2,1024,4,512),(2,768,8,128),(2,384,16,64),(2,1024,2,512),(2,1024,1,1024),(32,1024,16,256)]:
    x = torch.randn(n, ci, h, h, generator=g); w = torch.randn(ci, co, 4, 4, generator=g) * (2.0/(ci*16))**0.5
    ref = ref_conv(x, w, None, 0, 1)
    out = run_conv(ctx, 0, 1, 0, False, x, w, None, 0, ref.shape)
    chk = run_conv(ctx, 0, 1, 0, True, x, w, None, 0, ref.shape)
    print("convT", n, ci, h, co, "tiled", rel(out, ref), "naive", rel(chk, ref))
