"""The HIP path against the REAL reference at the resolutions BASELINE.json names (VERDICT r05 missing #2 / weak #1).

tests/golden/warp_step_256.npz     C2's resolution, 256 x 256, bs 2: resblocks on 16 x 16 maps, PatchGAN on 31 x 31
tests/golden/texture_step_256.npz  C3's resolution, 256 x 256, bs 1, 12 ROIs incl. a degenerate box, L1 + content + style on:
                                   the depth-8 U-Net of modules/swapnet_modules.py:176-190 + modules/pix2pix_modules.py:113-177
tests/golden/warp_step_c1.npz      C1: 64 x 64, bs 4, two steps

All three recorded by oracle/make_golden.py from models/{warp,texture}_model.py run on the CPU in the build container.  NO oracle
evaluation happens in these tests: weights come from the seed (the init functions are pinned to the reference's by
tests/test_oracle_golden.py), inputs from the seeded synthetic batch, and every comparison is library <-> reference.

Tolerances (north_star: 1e-3 relative fp32): losses 1e-3; fakes 1e-3 (24 samples + norm, and every stored element);
post-step weights 1e-3 on the norm, samples by _post_step_samples (Adam's first step is lr * sign(g): exact where the gradient is solid);
gradients: un-pinned, two fp32 evaluations (reference CPU / library) against each other, hence the 256 x 256 un-pinned bar of
tests/backends.assert_grads_vs_fp64 -- 3e-3 (2e-3 at 64 x 64) on the norm of every tensor, its 24 samples within that of their own value
plus 6 x that of the tensor's rms (single elements of a heavy-tailed error; at most 3 per tensor and 1 % overall inside the sign-flip
allowance of _gradient_samples, count printed), and
3e-3 rel-L2 with the sign-flip signature on the tensors stored whole (both ends of the backward chain).  Biases that feed an InstanceNorm are excluded (true gradient 0).
"""
import os

import numpy as np
import pytest
import torch

from oracle import swapnet_oracle as O
from oracle.golden_io import compare, compare_full, FULL_TENSORS
from swapnet_amd import engine
from tests import backends
from tests.test_texture_step import noise_bias as tex_noise_bias, vgg_state_dict
from tests.test_warp_step import noise_bias as warp_noise_bias

GPU = pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)
SIM = pytest.param("sim", id="hostsim")


def _ctx(kind):
    return backends.gpu_ctx() if kind == "gpu" else backends.hostsim_ctx()


def _check_step(gold, pre, full_set, L, out, gG, gD, pG, pD, skip, grad_tol):
    for key in [k for k in gold.files if k.startswith(pre + "loss/")]:
        ref = float(gold[key])
        k = key.rsplit("/", 1)[1]
        assert abs(L[k] - ref) <= 1e-3 * abs(ref) + 1e-6, (pre, k, L[k], ref, "vs reference")
    ok, msg = compare(gold, pre + "fakes", out, 1e-3, 1e-3)
    assert ok, msg
    n = soft = total = flipped = 0
    for grp, got in (("gradG", gG), ("gradD", gD)):
        for k, t in got.items():
            if skip(k) or (pre + grp + "/" + k + "/norm") not in gold.files:
                continue
            flipped += _gradient_samples(gold, pre + grp + "/" + k, t, grad_tol)
            n += 1
    print("%sgradient samples outside the diffuse bar but inside the sign-flip allowance: %d of %d" % (pre, flipped, 24 * n))
    assert flipped <= 0.01 * 24 * n, (flipped, n)
    for grp, got, grads, lr in (("postG", pG, gG, 1e-4), ("postD", pD, gD, 4e-4)):
        for k, t in got.items():
            if skip(k) or (pre + grp + "/" + k + "/norm") not in gold.files:
                continue
            a, b = _post_step_samples(gold, pre + grp + "/" + k, t, grads[k], lr)
            soft, total, n = soft + a, total + b, n + 1
    print("%s%d tensors against the reference; post-step samples: %d of %d on elements whose gradient is round-off-sized "
          "(|g| < 1e-2 rms: Adam's first step gives them an arbitrary sign in the reference too), held to 2 lr only" % (pre, n, soft, total))
    assert n > 50 and soft <= 0.15 * total, (n, soft, total)
    got = {"fakes": out, "gradG": gG, "gradD": gD, "postG": pG, "postD": pD}
    for key in FULL_TENSORS[full_set]:
        if not key.startswith(pre):
            continue
        grp, _, name = key[len(pre):].partition("/")
        t = got[grp] if grp == "fakes" else got[grp][name]
        ok, msg = compare_full(gold, key, t, rtol=1e-3 if grp == "fakes" else (3e-3 if grp.startswith("grad") else 1e-2),
                               flip_slices=3 if grp.startswith("grad") else 0,
                               outlier_frac=2e-3 if (grp.startswith("grad") and full_set.endswith("_256")) else 0.0)
        assert ok, msg
        if grp.startswith("grad"):
            print(msg)


def _gradient_samples(gold, key, t, tol):
    """Un-pinned gradient tensor against the reference's summary: the norm to `tol`; the 24 samples to tol x |ref| + 6 tol x rms (the
    error of an un-pinned gradient is heavy-tailed: a handful of branch flips, each spread over a layer's channels upstream of it;
    two kernel builds that round differently flip different elements) -- except that up to 3 of a tensor's 24 may sit outside it by up to 0.1 rms: one LeakyReLU /
    ReLU branch taken differently by the two fp32 evaluations moves ONE output channel of a layer's weight gradient by O(1) of that
    channel (tests/backends.assert_grads_vs_fp64), and a sample that lands in it sees that, not a diffuse error.  Returns the number
    of such samples (the caller prints the total and bounds it at 1 % of all samples)."""
    from oracle.golden_io import sample_idx
    t = t.detach().double().cpu().reshape(-1)
    gn = float(gold[key + "/norm"])
    assert abs(float(t.norm()) - gn) <= tol * gn, (key, "norm", float(t.norm()), gn)
    ref = np.asarray(gold[key + "/samples"], dtype=np.float64)
    err = np.abs(t[torch.from_numpy(sample_idx(t.numel(), key))].numpy() - ref)
    rms = gn / max(t.numel() ** 0.5, 1.0)
    out = err > tol * np.abs(ref) + 6 * tol * rms
    assert out.sum() <= 3 and bool(np.all(err[out] <= 0.1 * rms)), (
        key, "%d samples outside the bar, worst %.3e (rms %.3e, tol there %.3e)" % (out.sum(), err.max(), rms, (tol * np.abs(ref) + 6 * tol * rms)[err.argmax()]))
    return int(out.sum())


def _post_step_samples(gold, key, t, g, lr):
    """Post-step weights against the reference's summary.  Adam's FIRST update is lr * g / (|g| + 1e-8) = lr * sign(g): where the
    library's own gradient is solid (|g| >= 1e-2 rms(g)) the 24 sampled weights must match to 1e-3 of their value +
    0.05 lr (a wrong sign would be 2 lr); elsewhere a sign flip is legitimate and the element is held to 2.1 lr.  The norm to 1e-3.
    Returns (samples on the soft bar, samples)."""
    from oracle.golden_io import sample_idx
    t, g = t.detach().double().cpu().reshape(-1), g.detach().double().cpu().reshape(-1)
    gn = float(gold[key + "/norm"])
    assert abs(float(t.norm()) - gn) <= 1e-3 * gn, (key, float(t.norm()), gn)
    idx = torch.from_numpy(sample_idx(t.numel(), key))
    ref = torch.from_numpy(np.asarray(gold[key + "/samples"], dtype=np.float64))
    solid = g[idx].abs() >= 1e-2 * float(g.norm()) / max(t.numel() ** 0.5, 1.0)
    tol = torch.where(solid, 1e-3 * ref.abs() + 0.05 * lr, torch.full_like(ref, 2.1 * lr) + 1e-3 * ref.abs())
    err = (t[idx] - ref).abs()
    assert bool((err <= tol).all()), (key, "worst sample error %.3e (tol %.3e, solid %s)" % (
        float(err.max()), float(tol[err.argmax()]), bool(solid[err.argmax()])))
    return int((~solid).sum()), int(solid.numel())


def _warp_against_reference(backend, gold, full_set, grad_tol, assert_route=False):
    ctx = _ctx(backend)
    B, H = int(gold["meta/B"]), int(gold["meta/H"])
    torch.manual_seed(int(gold["meta/init_seed"]))
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=1234)
    m = engine.NativeModel(ctx, "warp", B, H, H)
    try:
        backends.reset_state(m, {engine.NET_G: G, engine.NET_D: D})
        for i, t in enumerate(batch):
            m.set_input(i, t)
        for si in range(len(gold["meta/step_seeds"])):
            pre = "step%d/" % si
            lab = [float(x) for x in gold[pre + "labels"]]
            if si == 0:
                m.forward(False, 0)
                for key in [k for k in gold.files if k.startswith("fwd/") and k.endswith("/norm")]:      # the 19 level taps of the reference's forward hooks
                    name = key[4:-5]
                    tap = {"body_down": "body_d", "cloth_down": "cloth_d", "cloth_up": "cloth_u", "resblocks.": "res", "dual_up": "dual_u"}
                    mine = next(v + name[len(k):] for k, v in tap.items() if name.startswith(k))
                    t = m.tap(engine.NET_G, mine)
                    nch = {"body_d1": 64, "cloth_d1": 64}.get(mine, t.shape[1])
                    ok, msg = compare(gold, "fwd/" + name, t[:, :nch], 1e-3, 1e-3)
                    assert ok, msg
            else:
                continue           # a free-running second step compares trajectories (tests/test_warp_step.py); step 0 is the pin
            m.backward_D(lab[0], lab[1])
            gD = m.state_dict(engine.NET_D, which=engine.W_GRAD, to_cpu=True)
            m.optimizer_step(engine.NET_D)
            m.backward_G(lab[2])
            gG = m.state_dict(engine.NET_G, which=engine.W_GRAD, to_cpu=True)
            m.optimizer_step(engine.NET_G)
            _check_step(gold, pre, full_set, m.losses(), m.output(), gG, gD, m.state_dict(engine.NET_G, to_cpu=True),
                        m.state_dict(engine.NET_D, to_cpu=True), warp_noise_bias, grad_tol)
    finally:
        m.close()


@pytest.mark.gpu
def test_warp_step_at_256_matches_the_reference(golden_dir):
    """BASELINE.json C2's resolution under the DEFAULT kernel routing (no marker: the product's thresholds)."""
    _warp_against_reference("gpu", np.load(os.path.join(golden_dir, "warp_step_256.npz")), "warp_256", 3e-3)


@pytest.mark.parametrize("backend", [SIM, GPU])
def test_warp_step_c1_matches_the_reference(backend, golden_dir):
    """BASELINE.json C1 (64 x 64, bs 4) under the default routing."""
    _warp_against_reference(backend, np.load(os.path.join(golden_dir, "warp_step_c1.npz")), "warp_c1", 2e-3)


@pytest.mark.gpu
def test_texture_step_at_256_matches_the_reference(golden_dir):
    """BASELINE.json C3's resolution: depth-8 U-Net, RoIAlign 256 -> 128 (bit-exact tap), VGG16 at 256 x 256."""
    gold = np.load(os.path.join(golden_dir, "texture_step_256.npz"))
    ctx = backends.gpu_ctx()
    B, H = int(gold["meta/B"]), int(gold["meta/H"])
    torch.manual_seed(int(gold["meta/init_seed"]))
    G, D = O.texture_module_params(img_size=H), O.patchgan_params(22)
    vgg = O.vgg16_feature_params()
    assert sum(1 for k in G if k.endswith(".weight") and k.startswith("unet.")) == 16          # 8 down + 8 up convs
    batch = O.synth_texture_batch(B, H, H, seed=4321)
    m = engine.NativeModel(ctx, "texture", B, H, H, is_train=True)
    try:
        assert list(m.param_infos(engine.NET_G).keys()) == list(G.keys())
        m.load_state_dict(engine.NET_G, G)
        m.load_state_dict(engine.NET_D, D)
        m.load_state_dict(engine.NET_VGG, vgg_state_dict(m, vgg))
        m.set_hyper()
        for i, t in enumerate(batch):
            m.set_input(i, t)
        lab = [float(x) for x in gold["step0/labels"]]
        m.forward(False, 0)
        ok, msg = compare(gold, "fwd/roi_align", m.tap(engine.NET_G, "pooled").reshape(B * 12, 3, 128, 128), 1e-6, 1e-6)
        assert ok, msg
        ok, msg = compare(gold, "fwd/encode", m.tap(engine.NET_G, "encoded")[:, :36], 1e-3, 1e-3)
        assert ok, msg
        m.backward_D(lab[0], lab[1])
        gD = m.state_dict(engine.NET_D, which=engine.W_GRAD, to_cpu=True)
        m.optimizer_step(engine.NET_D)
        m.backward_G(lab[2])
        gG = m.state_dict(engine.NET_G, which=engine.W_GRAD, to_cpu=True)
        m.optimizer_step(engine.NET_G)
        keys = list(G.keys())
        _check_step(gold, "step0/", "texture_256", m.losses(), m.output(), gG, gD, m.state_dict(engine.NET_G, to_cpu=True),
                    m.state_dict(engine.NET_D, to_cpu=True), lambda k: tex_noise_bias(k, keys), 3e-3)
    finally:
        m.close()
