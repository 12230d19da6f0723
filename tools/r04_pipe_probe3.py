"""Where does call 3 (replay, inputs 1) first differ from call 1 (eager, inputs 1)?  Texture-stage taps + output, repeated
until a mismatch shows (round 4: flaky test_two_stage_device_pipeline_*)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import swapnet_oracle as O
from swapnet_amd import engine
from swapnet_amd.pipeline import TwoStagePipeline

ctx = engine.Context(workspace_mb=1024)
torch.manual_seed(3)
Gw, Gt = O.warp_module_params(), O.texture_module_params(img_size=64)
b1, i1, _ = O.synth_warp_batch(1, 64, 64, seed=9)
t1, r1, _, _ = O.synth_texture_batch(1, 64, 64, seed=10)
b2, i2, _ = O.synth_warp_batch(1, 64, 64, seed=19)
t2, r2, _, _ = O.synth_texture_batch(1, 64, 64, seed=20)
with torch.no_grad():
    warped = O.warp_module_forward(Gw, b1, i1)
    cloth = O.labels_to_onehot(O.onehot_to_labels(warped), 19)
    ref = O.texture_module_forward(Gt, t1, r1, cloth)

def taps(pipe):
    t = pipe.texture.cur
    out = {k: t.tap(engine.NET_G, k).clone() for k in ("pooled", "encoded", "unet_in", "fakes")}
    w = pipe.warp.cur
    out["warp_out"] = w.output().clone()
    return out

for trial in range(int(os.environ.get("PROBE_TRIALS", "2"))):
    pipe = TwoStagePipeline(Gw, Gt, img_size=64, ctx=ctx)
    o1, l1 = pipe(b1, i1, t1, r1, return_labels=True); o1 = o1.clone(); T1 = taps(pipe)
    time.sleep(0.5)
    o2, l2 = pipe(b2, i2, t2, r2, return_labels=True)
    time.sleep(0.5)
    if len(sys.argv) > 1:        # a second pipeline in the same context, run eagerly between the replays (what the test does)
        eager = TwoStagePipeline(Gw, Gt, img_size=64, ctx=ctx, use_graph=False)
        o2e, l2e = eager(b2, i2, t2, r2, return_labels=True)
        print("   eager2 vs replay2 max|d| %.3e" % float((o2 - o2e).abs().max()))
        if sys.argv[1] == "sync":
            torch.cuda.synchronize(); ctx.sync()
    o3, l3 = pipe(b1, i1, t1, r1, return_labels=True); o3 = o3.clone(); T3 = taps(pipe)
    d = {k: float((T1[k] - T3[k]).abs().max()) for k in T1}
    e1 = float((o1.cpu() - ref).norm() / ref.norm()); e3 = float((o3.cpu() - ref).norm() / ref.norm())
    print("trial %d out max|d| %.3e  err vs oracle: call1 %.2e call3 %.2e  taps %s" % (trial, float((o1 - o3).abs().max()), e1, e3,
          " ".join("%s %.1e" % kv for kv in d.items())), flush=True)
    del pipe
