"""test_two_stage_device_pipeline_* scenario with a second pipeline in the same context between the replays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import swapnet_oracle as O
from swapnet_amd import engine
from swapnet_amd.pipeline import TwoStagePipeline

mode = sys.argv[1] if len(sys.argv) > 1 else "run"
ctx = engine.Context(workspace_mb=1024)
torch.manual_seed(3)
Gw, Gt = O.warp_module_params(), O.texture_module_params(img_size=64)
b1, i1, _ = O.synth_warp_batch(1, 64, 64, seed=9)
t1, r1, _, _ = O.synth_texture_batch(1, 64, 64, seed=10)
b2, i2, _ = O.synth_warp_batch(1, 64, 64, seed=19)
t2, r2, _, _ = O.synth_texture_batch(1, 64, 64, seed=20)
pipe = TwoStagePipeline(Gw, Gt, img_size=64, ctx=ctx)
o1, l1 = pipe(b1, i1, t1, r1, return_labels=True); o1, l1 = o1.clone(), l1.clone()
o2, l2 = pipe(b2, i2, t2, r2, return_labels=True); o2, l2 = o2.clone(), l2.clone()
if mode != "none":
    eager = TwoStagePipeline(Gw, Gt, img_size=64, ctx=ctx, use_graph=(mode == "graph"))
    if mode != "create":
        o2e, l2e = eager(b2, i2, t2, r2, return_labels=True)
        print("eager2 vs replay2: labels differ %d out max|d| %.3e" % (int((l2 != l2e).sum()), float((o2 - o2e).abs().max())))
o3, l3 = pipe(b1, i1, t1, r1, return_labels=True); o3, l3 = o3.clone(), l3.clone()
o4, l4 = pipe(b1, i1, t1, r1, return_labels=True)
print("mode %-6s | call1 vs call3: labels differ %d, out max|d| %.3e | call3 vs call4: labels differ %d, out max|d| %.3e" % (
    mode, int((l1 != l3).sum()), float((o1 - o3).abs().max()), int((l3 != l4).sum()), float((o3 - o4).abs().max())), flush=True)
wt = pipe.warp.cur
print("warp model shapes:", list(pipe.warp.models.keys()), " texture:", list(pipe.texture.models.keys()))
