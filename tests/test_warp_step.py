"""Warp-stage parity: the native G+D step
against (i) the CPU oracle on the same seeded inputs and (ii) the golden vectors recorded
from the real reference (tests/golden/warp_step_64.npz, config C1 shape 64x64).

fp32 tolerances (north_star: 1e-3 relative):
  forward activations / fakes / losses : 1e-3 (observed ~1e-6 .. 1e-5)
  post-step weights                    : 1e-3 rel-L2 per tensor
  gradients                            : measured against the SAME step evaluated in float64 (oracle dtype=float64):
    rel-L2(native, fp64) <= max(1e-3, 2 x rel-L2(torch fp32, fp64)) per tensor (backends.assert_grads_vs_fp64).
    The reference's own fp32 CPU backward is ~1.3e-3 away from the fp64 evaluation on the deep layers
    (printed by tests/test_train_parity.py::test_gradients_against_fp64_oracle), so a fixed 1e-3 against the
    fp32 oracle would test the oracle's round-off, not our kernels.
Biases that feed an InstanceNorm are excluded from grad / post-step checks: their true
gradient is 0, the reference's is round-off noise that Adam normalises to +-lr.
"""
import os

import numpy as np
import pytest
import torch

from oracle import swapnet_oracle as O
from oracle.golden_io import compare, compare_full, FULL_TENSORS
from swapnet_amd import engine
from tests import backends

BACKENDS = [pytest.param("sim", id="hostsim"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _ctx(kind):
    return backends.gpu_ctx() if kind == "gpu" else backends.hostsim_ctx()


def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))


def noise_bias(name):
    return name.endswith(".bias") and ("resblocks" in name or name.startswith(("model.2.", "model.5.", "model.8.")))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "warp_step_64.npz"))


@pytest.fixture(scope="module")
def oracle_run(gold):
    torch.manual_seed(int(gold["meta/init_seed"]))
    G, D = O.warp_module_params(), O.patchgan_params(22)
    B, H = int(gold["meta/B"]), int(gold["meta/H"])
    batch = O.synth_warp_batch(B, H, H, seed=1234)
    taps = {}
    with torch.no_grad():
        O.warp_module_forward(G, *batch[:2], taps=taps)
    st = O.WarpStepOracle(G, D)
    steps = []
    for seed in gold["meta/step_seeds"]:
        s64 = st.astype(torch.float64)          # same pre-step state, evaluated in double
        torch.manual_seed(int(seed))
        st.step(*batch)
        s64.step(*batch, labels=st.labels)
        steps.append(dict(losses=dict(st.losses), labels=list(st.labels), fakes=st.fakes.clone(),
                          g64G=s64.grads_G, g64D=s64.grads_D,
                          gG={k: v.clone() for k, v in st.grads_G.items()},
                          gD={k: v.clone() for k, v in st.grads_D.items()},
                          pG={k: v.clone() for k, v in st.G.items()}, pD={k: v.clone() for k, v in st.D.items()},
                          mG={k: v.clone() for k, v in st.optG.m.items()}, vG={k: v.clone() for k, v in st.optG.v.items()},
                          mD={k: v.clone() for k, v in st.optD.m.items()}, vD={k: v.clone() for k, v in st.optD.v.items()}))
    return G, D, batch, taps, steps


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_warp_forward_levels(backend, oracle_run, gold):
    G, D, batch, taps, _ = oracle_run
    ctx = _ctx(backend)
    B, H = batch[0].shape[0], batch[0].shape[2]
    m = backends.get_model(ctx, "warp", B, H, is_train=False)
    backends.reset_state(m, {engine.NET_G: G})
    m.set_input(0, batch[0]); m.set_input(1, batch[1])
    m.forward(False, 0)
    for name, ref in taps.items():
        t = m.tap(engine.NET_G, name)
        assert rel(t[:, :ref.shape[1]], ref) < 1e-3, (name, rel(t[:, :ref.shape[1]], ref))
    out = m.output()
    ok, msg = compare(gold, "step0/fakes", out, rtol=1e-3, atol_frac=1e-3)
    assert ok, msg
    # state_dict round trip is exact (pack -> unpack)
    sd = m.state_dict(engine.NET_G, to_cpu=True)
    assert list(sd.keys()) == list(G.keys())
    for k in G:
        assert torch.equal(sd[k], G[k]), k


def _accept_only_an_evidenced_sign_flip(m, G0, D0, steps, si, batch, gD, gG, unpinned):
    """The un-pinned bar failed.  The one legitimate cause is a LeakyReLU / ReLU branch taken differently for a
    pre-activation within round-off of zero: the element's layer sees it in one output channel (what the slice exemption
    of assert_grads_vs_fp64 covers), but every layer UPSTREAM receives it spread over all channels (observed on the
    MI355X in round 4: PatchGAN model.0 weight / bias 1.5e-3 at step 1 from one flip in model.1's 16x16 map, identical
    under six kernel-routing switches, gone under the two that change model.1's rounding).  So the excuse must be
    EVIDENCED, not assumed: replay the native pass's branch pattern in the float64 oracle started from the same
    pre-step state; accept only if (a) the replay counts at least one branch that differs from the oracle's own and
    (b) with the pattern pinned every gradient tensor is within 1e-4 of float64.  No flip, or a pinned mismatch,
    re-raises the original failure."""
    from collections import OrderedDict
    if si == 0:
        o = O.WarpStepOracle(G0, D0, dtype=torch.float64)
    else:
        p = steps[si - 1]
        o = O.WarpStepOracle(p["pG"], p["pD"], dtype=torch.float64)
        for opt, mk, vk in ((o.optG, "mG", "vG"), (o.optD, "mD", "vD")):
            opt.m = OrderedDict((k, v.double().clone()) for k, v in p[mk].items())
            opt.v = OrderedDict((k, v.double().clone()) for k, v in p[vk].items())
            opt.step = si
    replay = O.PatternReplay(backends.collect_patterns(m))
    o.patterns = replay
    o.step(*batch, labels=steps[si]["labels"])
    flips = replay.check()
    if sum(flips.values()) == 0 and unpinned.args != ("forced",):
        raise unpinned
    try:
        wD = backends.assert_grads_replayed(gD, o.grads_D, noise_bias, 1e-4, (si, "gradD pinned"))
        wG = backends.assert_grads_replayed(gG, o.grads_G, noise_bias, 1e-4, (si, "gradG pinned"))
    except AssertionError as pinned:
        raise AssertionError((unpinned.args, "and with the pattern pinned:", pinned.args))
    print("step %d: un-pinned bar missed (%s); %s branch flips against the float64 oracle, pinned gradients within %.1e (D) / %.1e (G)"
          % (si, str(unpinned.args)[:200], flips, wD, wG))


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_warp_two_steps_match_oracle_and_reference(backend, oracle_run, gold):
    G, D, batch, _, steps = oracle_run
    ctx = _ctx(backend)
    B, H = batch[0].shape[0], batch[0].shape[2]
    m = backends.get_model(ctx, "warp", B, H)
    backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
    for i, t in enumerate(batch):
        m.set_input(i, t)
    for si, s in enumerate(steps):
        pre = "step%d/" % si
        if si > 0:
            # Adam's first update is +-lr * sign(g): round-off-level gradient differences flip
            # signs of near-zero entries and GAN dynamics amplify that chaotically, so a free-
            # running second step would compare trajectories, not kernels.  Re-synchronise the
            # full training state (weights + both Adam moments + step count) with the oracle's
            # and check the step>1 path (bias corrections, moment EMA) from identical state.
            p = steps[si - 1]
            for net, wk, mk, vk in ((engine.NET_G, "pG", "mG", "vG"), (engine.NET_D, "pD", "mD", "vD")):
                m.load_state_dict(net, p[wk])
                m.load_state_dict(net, p[mk], which=engine.W_EXP_AVG)
                m.load_state_dict(net, p[vk], which=engine.W_EXP_AVG_SQ)
                m.optim_step_count(net, si)
        # phase by phase, in the reference's order (models/base_gan.py:194-203)
        m.forward(False, 0)
        m.backward_D(s["labels"][0], s["labels"][1])
        gD = m.state_dict(engine.NET_D, which=engine.W_GRAD, to_cpu=True)
        m.optimizer_step(engine.NET_D)
        m.backward_G(s["labels"][2])
        gG = m.state_dict(engine.NET_G, which=engine.W_GRAD, to_cpu=True)
        m.optimizer_step(engine.NET_G)
        L = m.losses()
        for k, v in s["losses"].items():
            assert abs(L[k] - v) <= 1e-3 * abs(v) + 1e-6, (si, k, L[k], v)
            assert abs(L[k] - float(gold[pre + "loss/" + k])) <= 1e-3 * abs(v) + 1e-6, (si, k, "vs reference")
        out = m.output()
        assert rel(out, s["fakes"]) < 1e-3
        ok, msg = compare(gold, pre + "fakes", out, 1e-3, 1e-3)
        assert ok, msg
        try:
            if os.environ.get("SWAPNET_TEST_FORCE_PINNED") == "1":        # (exercises the fall-back below where nothing flips)
                raise AssertionError("forced")
            backends.assert_grads_vs_fp64(gD, s["gD"], s["g64D"], noise_bias, (si, "gradD"))
            backends.assert_grads_vs_fp64(gG, s["gG"], s["g64G"], noise_bias, (si, "gradG"))
        except AssertionError as unpinned:
            _accept_only_an_evidenced_sign_flip(m, G, D, steps, si, batch, gD, gG, unpinned)
        pG = m.state_dict(engine.NET_G, to_cpu=True)
        pD = m.state_dict(engine.NET_D, to_cpu=True)
        for k, v in s["pG"].items():
            if not noise_bias(k):
                assert rel(pG[k], v) < 1e-3, (si, "postG", k, rel(pG[k], v))
                ok, msg = compare(gold, pre + "postG/" + k, pG[k], 1e-3, 3e-3)
                assert ok, msg
        for k, v in s["pD"].items():
            if not noise_bias(k):
                assert rel(pD[k], v) < 1e-3, (si, "postD", k, rel(pD[k], v))
        # the tensors the golden file stores WHOLE (recorded from the real reference): every element of the
        # generator output and of the gradients at both ends of the backward chain, not 24 samples
        got = {"fakes": out, "gradG": gG, "gradD": gD, "postG": pG, "postD": pD}
        for key in FULL_TENSORS["warp"]:
            if key.startswith(pre):
                grp, _, name = key[len(pre):].partition("/")
                t = got[grp] if grp == "fakes" else got[grp][name]
                ok, msg = compare_full(gold, key, t, rtol=1e-3 if grp == "fakes" else (3e-3 if grp.startswith("grad") else 1e-2),
                                   flip_slices=3 if grp.startswith("grad") else 0)
                assert ok, msg
    # Adam moments come back in the reference's state-dict layout
    ea = m.state_dict(engine.NET_G, which=engine.W_EXP_AVG, to_cpu=True)
    assert ea["body_down1.model.0.weight"].shape == (64, 3, 4, 4)
    assert m.optim_step_count(engine.NET_G) == len(steps)


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_step_equals_phased_step(backend, oracle_run):
    """swn_model_step == forward; backward_D; step D; backward_G; step G (bitwise)."""
    G, D, batch, _, steps = oracle_run
    ctx = _ctx(backend)
    B, H = batch[0].shape[0], batch[0].shape[2]
    outs = []
    for fused in (False, True):
        m = backends.get_model(ctx, "warp", B, H)
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        lab = steps[0]["labels"]
        if fused:
            m.step(lab, training=False, seed=0)
        else:
            m.forward(False, 0); m.backward_D(lab[0], lab[1]); m.optimizer_step(1); m.backward_G(lab[2]); m.optimizer_step(0)
        outs.append((m.losses(), m.state_dict(0, to_cpu=True)["upsample_and_pad.2.weight"]))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_early_cross_entropy_term_is_the_same_step(backend, oracle_run, monkeypatch):
    """Round 6 takes loss_G's cross-entropy term early (behind the discriminator's backward pass; PatchGAN's input gradient accumulates onto
    it) instead of accumulating it onto PatchGAN's gradient inside backward_G (warp_model.py:141-167).  a + b = b + a: every gradient of
    both networks, the losses and the post-step weights are BIT-identical between the two orders, phase by phase, and also when backward_G
    runs without a backward_D in front of it (the early term is then taken inside backward_G)."""
    G, D, batch, _, steps = oracle_run
    ctx = _ctx(backend)
    B, H = batch[0].shape[0], batch[0].shape[2]
    lab = steps[0]["labels"]
    outs = []
    for early in ("1", "0"):
        monkeypatch.setenv("SWN_CE_EARLY", early)                    # read when a model is built
        m = engine.NativeModel(ctx, "warp", B, H, H)
        try:
            backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
            for i, t in enumerate(batch):
                m.set_input(i, t)
            m.forward(False, 0); m.backward_G(lab[2])                 # no backward_D in front
            g_alone = m.grad_arena(engine.NET_G).clone()
            m.forward(False, 0); m.backward_D(lab[0], lab[1]); m.optimizer_step(1); m.backward_G(lab[2])
            gG, gD = m.grad_arena(engine.NET_G).clone(), m.grad_arena(engine.NET_D).clone()
            m.optimizer_step(0)
            m.ctx.sync()
            outs.append((m.losses(), g_alone.cpu(), gG.cpu(), gD.cpu(), m.weight_arena(engine.NET_G).clone().cpu()))
        finally:
            m.close()
    a, b = outs
    assert a[0] == b[0], (a[0], b[0])
    for x, y, what in zip(a[1:], b[1:], ("G gradients, backward_G alone", "G gradients", "D gradients", "G weights after the step")):
        assert torch.equal(x, y), what
    assert float(a[2].abs().max()) > 0


@pytest.mark.gpu
def test_warp_full_size_properties():
    """Config C2 shape (256x256, bs 32): size-independent properties the oracle cannot reach.
    (1) run-to-run bitwise determinism of a full training step (fixed-order reductions);
    (2) batch-permutation equivariance of the generator (every layer is per-sample), to summation-order round-off;
    (3) dropout: train-mode forward differs from eval, and is reproducible for a fixed seed."""
    ctx = backends.gpu_ctx()
    B, H = 32, 256
    torch.manual_seed(0)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=1234)
    m = engine.NativeModel(ctx, "warp", B, H, H, is_train=True)
    res = []
    for _ in range(2):
        m.load_state_dict(0, G); m.load_state_dict(1, D); m.set_hyper()
        for w in (engine.W_EXP_AVG, engine.W_EXP_AVG_SQ):
            m.load_state_dict(0, {k: torch.zeros_like(v) for k, v in G.items()}, which=w)
            m.load_state_dict(1, {k: torch.zeros_like(v) for k, v in D.items()}, which=w)
        m.optim_step_count(0, 0); m.optim_step_count(1, 0)
        for i, t in enumerate(batch):
            m.set_input(i, t)
        m.step([0.9, 0.8, 1.0], training=True, seed=7)
        res.append((m.losses(), m.output().cpu(), m.weight_arena(0).clone().cpu()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert all(np.isfinite(v) for v in res[0][0].values())
    # (2) permutation equivariance in eval mode
    m.forward(False, 0)
    a = m.output().cpu()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    m.set_input(0, batch[0][perm]); m.set_input(1, batch[1][perm])
    m.forward(False, 0)
    b = m.output().cpu()
    # (not bitwise: the tile a sample falls into decides whether its K reduction runs whole or as the split tail
    # of the launch -- a different, still fixed, summation order; measured 2.1e-6 rel-L2, the forward pass's own
    # fp32 round-off against float64 being 1-4e-6)
    assert float((a[perm] - b).abs().max()) < 1e-4 and rel(b, a[perm]) < 1e-5
    # (3) dropout
    m.forward(True, 5); d1 = m.output().cpu()
    m.forward(True, 5); d2 = m.output().cpu()
    m.forward(True, 6); d3 = m.output().cpu()
    assert torch.equal(d1, d2) and not torch.equal(d1, d3) and not torch.equal(d1, b)
    assert a.abs().max() <= 1.0
    m.close()


@pytest.mark.gpu
def test_warp_step_at_full_resolution_matches_oracle():
    """One full G+D step at the C2 resolution (256x256; bs 2 so the CPU oracle finishes in seconds): the
    shapes every full-size kernel path sees -- F(4x4,3x3) on 16x16 maps with the 18x18 reflect-padded
    gradient, F(3x3,4x4) on PatchGAN's 31x31 map, the fused 4-phase tail conv at 128x128, the head conv --
    against the oracle: losses / fakes 1e-3, gradients by the fp64 yardstick (module docstring)."""
    ctx = backends.gpu_ctx()
    B, H = 2, 256
    torch.manual_seed(3)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=99)
    st = O.WarpStepOracle({k: v.clone() for k, v in G.items()}, {k: v.clone() for k, v in D.items()})
    s64 = st.astype(torch.float64)
    torch.manual_seed(17)
    st.step(*batch)
    s64.step(*batch, labels=st.labels)
    m = engine.NativeModel(ctx, "warp", B, H, H)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        m.forward(False, 0)
        m.backward_D(st.labels[0], st.labels[1])
        gD = m.state_dict(engine.NET_D, which=engine.W_GRAD, to_cpu=True)
        m.optimizer_step(engine.NET_D)
        m.backward_G(st.labels[2])
        gG = m.state_dict(engine.NET_G, which=engine.W_GRAD, to_cpu=True)
        L = m.losses()
        for k, v in st.losses.items():
            assert abs(L[k] - v) <= 1e-3 * abs(v) + 1e-6, (k, L[k], v)
        assert rel(m.output(), st.fakes) < 1e-3
        # un-pinned at 256x256: sign-flip noise floor (backends.assert_grads_vs_fp64); pinned: test_pattern_replay.py
        backends.assert_grads_vs_fp64(gD, st.grads_D, s64.grads_D, noise_bias, "gradD 256", floor=1e-3, mult=4.0, cap=5e-3)
        backends.assert_grads_vs_fp64(gG, st.grads_G, s64.grads_G, noise_bias, "gradG 256", floor=1e-3, mult=4.0, cap=5e-3)
    finally:
        m.close()


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", ["lsgan", "wgan", "ce"])
def test_other_gan_objectives_and_ce_mode(backend, mode, oracle_run):
    """--gan_mode lsgan / wgan (modules/loss.py:55-62,117-128) and --warp_mode ce
    (warp_model.py:169-183) against the oracle (the non-gp part)."""
    G, D, batch, _, steps = oracle_run
    ctx = _ctx(backend)
    B, H = batch[0].shape[0], batch[0].shape[2]
    hyper = dict(gan_mode="vanilla" if mode == "ce" else mode, warp_mode="ce" if mode == "ce" else "gan")
    st = O.WarpStepOracle(G, D, hyper=hyper)
    lab = steps[0]["labels"]
    st.step(*batch, labels=lab)
    m = backends.get_model(ctx, "warp", B, H)
    backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
    m.set_hyper(gan_mode={"lsgan": 1, "wgan": 2, "ce": 0}[mode], warp_mode_ce=int(mode == "ce"))
    for i, t in enumerate(batch):
        m.set_input(i, t)
    m.step(lab, training=False, seed=0)
    L = m.losses()
    for k, v in st.losses.items():
        assert abs(L[k] - v) <= 1e-3 * abs(v) + 1e-6, (mode, k, L[k], v)
    # ... and against the real reference's step under this objective (same seeds => same labels)
    gm = np.load(os.path.join(os.path.dirname(__file__), "golden", "warp_modes_64.npz"))
    for key in gm.files:
        if key.startswith(mode + "/loss/"):
            k, ref = key.split("/")[-1], float(gm[key])
            assert abs(L[k] - ref) <= 1e-3 * abs(ref) + 1e-6, (mode, k, L[k], ref, "vs reference")
    ok, msg = compare(gm, mode + "/fakes", m.output(), 1e-3, 1e-3)
    assert ok, msg
    pG = m.state_dict(0, to_cpu=True)
    for k, v in st.G.items():
        if not noise_bias(k):
            assert rel(pG[k], v) < 1e-3, (mode, "postG", k, rel(pG[k], v))
    ok, msg = compare(gm, mode + "/postG/upsample_and_pad.2.weight", pG["upsample_and_pad.2.weight"], 1e-3, 3e-3)
    assert ok, msg
    if mode == "ce":      # the discriminator is untouched
        pD = m.state_dict(1, to_cpu=True)
        assert all(torch.equal(pD[k], D[k]) for k in D)
        assert m.optim_step_count(1) == 0


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
def test_non_square_warp_forward(backend, oracle_run):
    """DeepFashion-shaped inputs are 4:3 (BASELINE.json config C5: 256x192): the warp generator
    only needs H and W to be multiples of 64.  64x128 against the oracle."""
    G = oracle_run[0]
    ctx = _ctx(backend)
    g = torch.Generator().manual_seed(11)
    body = torch.randn(1, 3, 64, 128, generator=g)
    lab = torch.randint(0, 19, (1, 8, 16), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
    cloth = O.labels_to_onehot(lab, 19)
    with torch.no_grad():
        ref = O.warp_module_forward(G, body, cloth)
    m = engine.NativeModel(ctx, "warp", 1, 64, 128, is_train=False)
    m.load_state_dict(engine.NET_G, G)
    m.set_input(0, body); m.set_input(1, cloth)
    m.forward(False, 0)
    assert m.output().shape == (1, 19, 64, 128)
    assert rel(m.output(), ref) < 1e-3
    m.close()


@pytest.mark.small_channel_winograd
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("variant", ["winograd_f2x2", "direct"])
def test_resblock_conv_variants_match_oracle(backend, variant, oracle_run, monkeypatch):
    """The 3x3 ResidualBlock convs run as Winograd F(4x4,3x3) by default (the shared models of the
    tests above).  The F(2x2,3x3) path (H or W not a multiple of 4) and the direct implicit-GEMM path
    (small channel counts) must give the same step: forward, D and G gradients vs the oracle."""
    if variant == "direct":
        monkeypatch.setenv("SWN_WINOGRAD", "0")
    else:
        monkeypatch.setenv("SWN_WINO_M", "2")
    G, D, batch, _, steps = oracle_run
    ctx = _ctx(backend)
    B, H = batch[0].shape[0], batch[0].shape[2]
    m = engine.NativeModel(ctx, "warp", B, H, H)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        s = steps[0]
        m.forward(False, 0)
        assert rel(m.output(), s["fakes"]) < 1e-3
        m.backward_D(s["labels"][0], s["labels"][1])
        m.optimizer_step(engine.NET_D)
        m.backward_G(s["labels"][2])
        gG = m.state_dict(engine.NET_G, which=engine.W_GRAD, to_cpu=True)
        backends.assert_grads_vs_fp64(gG, s["gG"], s["g64G"], noise_bias, variant)
    finally:
        m.close()


@pytest.mark.gpu
def test_fast_algorithms_track_the_direct_kernels_over_training_steps(monkeypatch):
    """Winograd (F(4x4,3x3), F(3x3,4x4)), the fused tail kernels and the taps-on-N head change the
    summation order, not the math.  Five free-running training steps (dropout on, Adam) at 128x128 with
    all of them against five steps with every conv on the direct implicit-GEMM kernel: the loss
    trajectories must stay within 1 % (Adam's sign-like first steps amplify round-off, so this bounds
    drift, the per-step parity tests bound error)."""
    ctx = backends.gpu_ctx()
    B, H = 4, 128
    torch.manual_seed(2)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=21)
    traj = []
    for direct in (False, True):
        for k in ("SWN_WINOGRAD", "SWN_TAIL4", "SWN_HEAD_TAPN", "SWN_NARROW"):
            (monkeypatch.setenv(k, "0") if direct else monkeypatch.delenv(k, raising=False))
        m = engine.NativeModel(ctx, "warp", B, H, H, is_train=True)
        try:
            backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
            for i, t in enumerate(batch):
                m.set_input(i, t)
            steps = []
            for s in range(5):
                m.step([0.9, 0.8, 1.0], training=True, seed=100 + s)
                steps.append(m.losses())
            traj.append(steps)
        finally:
            m.close()
    for a, b in zip(*traj):
        for k in ("D", "G", "G_ce", "G_gan"):
            assert abs(a[k] - b[k]) <= 1e-2 * abs(b[k]) + 1e-4, (k, a[k], b[k])
