"""BASELINE.json C5's "hipGraph-captured step": swn_model_step_captured records the G+D optimize_parameters step
(models/base_gan.py:194-203) once and replays it; the per-step scalars -- GANLoss's three smooth-label draws
(modules/loss.py:77-104), the dropout seed, both AdamW bias corrections -- travel through device memory.  Whatever the form,
the step must be the SAME step: bit-identical weights, Adam moments and losses against swn_model_step over several steps with
different labels and seeds.  On the host simulator (no graphs) the test covers the device-memory parameter plumbing; on the
MI355X the recorded two-stream launch sequence itself."""
import pytest
import torch

from oracle import swapnet_oracle as O
from swapnet_amd import engine
from tests import backends

pytestmark = pytest.mark.small_channel_winograd

BACKENDS = [pytest.param("sim", id="hostsim"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _run(ctx, kind, B, H, captured, steps, training=True):
    torch.manual_seed(3)
    if kind == "warp":
        G, D = O.warp_module_params(), O.patchgan_params(22)
        batch = O.synth_warp_batch(B, H, H, seed=5)
    else:
        G, D = O.texture_module_params(img_size=H), O.patchgan_params(22)
        batch = O.synth_texture_batch(B, H, H, seed=5)
    m = engine.NativeModel(ctx, kind, B, H, H, is_train=True)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        out = []
        for k in range(steps):
            labels = [0.7 + 0.05 * k, 1.05 - 0.04 * k, 0.9 + 0.01 * k]
            m.step(labels, training=training, seed=100 + 17 * k, captured=captured)
            out.append((m.losses(), m.optim_step_count(engine.NET_G), m.optim_step_count(engine.NET_D)))
        state = [m.arena(net, which).clone().cpu() for net in (engine.NET_G, engine.NET_D)
                 for which in (engine.W_WEIGHT, engine.W_EXP_AVG, engine.W_EXP_AVG_SQ)]
        # the model keeps working eagerly afterwards (operand refresh follows the weights the replays updated)
        m.step([0.9, 0.8, 1.0], training=training, seed=7)
        state.append(m.arena(engine.NET_G, engine.W_WEIGHT).clone().cpu())
        return out, state
    finally:
        m.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("kind", ["warp", "texture"])
def test_captured_step_is_the_same_step(backend, kind):
    ctx = backends.gpu_ctx() if backend == "gpu" else backends.hostsim_ctx()
    B, H, steps = (2, 64, 3) if backend == "sim" else (4, 128, 5)
    eager = _run(ctx, kind, B, H, False, steps)
    capt = _run(ctx, kind, B, H, True, steps)
    for k, (a, b) in enumerate(zip(eager[0], capt[0])):
        assert a == b, (kind, "step", k, a, b)
    for i, (a, b) in enumerate(zip(eager[1], capt[1])):
        assert torch.equal(a, b), (kind, "state tensor", i, float((a - b).abs().max()))


@pytest.mark.gpu
def test_captured_step_at_c2_in_eval_and_train_mode():
    """BASELINE.json C2's shape (256x256, bs 32): one graph per mode, both bit-identical to the eager step."""
    ctx = backends.gpu_ctx()
    for training in (True, False):
        eager = _run(ctx, "warp", 32, 256, False, 3, training)
        capt = _run(ctx, "warp", 32, 256, True, 3, training)
        assert eager[0] == capt[0]
        assert all(torch.equal(a, b) for a, b in zip(eager[1], capt[1]))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("kind", ["warp", "texture"])
def test_adamw_streamed_behind_each_bucket_is_the_same_step(backend, kind, monkeypatch):
    """swn_model_step applies AdamW bucket by bucket behind each bucket's weight gradients (side stream, under the
    back-propagation of the earlier layers) instead of once after optimizer_G.step()'s position in
    models/base_gan.py:200-203.  Same element-wise update from the same gradients: weights, both moments, step counters
    and losses must be bit-identical to the single launch (SWN_STREAM_ADAMW=0).  The host simulator has no second
    stream; =2 forces the bucketed order there (what can go wrong on the CPU side is the bucket -> arena-range map)."""
    ctx = backends.gpu_ctx() if backend == "gpu" else backends.hostsim_ctx()
    B, H, steps = (2, 64, 3) if backend == "sim" else (4, 128, 4)
    monkeypatch.setenv("SWN_STREAM_ADAMW", "0")
    whole = _run(ctx, kind, B, H, False, steps)
    monkeypatch.setenv("SWN_STREAM_ADAMW", "2" if backend == "sim" else "1")
    bucketed = _run(ctx, kind, B, H, False, steps)
    captured = _run(ctx, kind, B, H, True, steps)
    for other in (bucketed, captured):
        assert whole[0] == other[0]
        for i, (a, b) in enumerate(zip(whole[1], other[1])):
            assert torch.equal(a, b), (kind, "state tensor", i, float((a - b).abs().max()))
