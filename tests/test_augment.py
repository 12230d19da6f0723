"""the warp dataloader's per-channel augmentation (datasets/data_utils.py:346-361) as one
device gather.  Oracle = Pillow itself (the library the reference calls through torchvision): every channel of every
sample is pushed through the SAME sequence of PIL operations with the SAME parameters and must come out bit-identical."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import swapnet_oracle as O
from swapnet_amd.datasets import gpu_augment as A
from tests import backends

BACKENDS = [pytest.param("sim", id="hostsim"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _ctx(kind):
    return backends.gpu_ctx() if kind == "gpu" else backends.hostsim_ctx()


def pil_chain(channel, chain_desc):
    """channel: (H, W) float32; chain_desc: list of ("hflip",) / ("vflip",) / ("affine", coeffs6) / ("persp", coeffs8)."""
    img = Image.fromarray(channel)
    for step in chain_desc:
        if step[0] == "hflip":
            img = img.transpose(Image.FLIP_LEFT_RIGHT)                 # torchvision F.hflip
        elif step[0] == "vflip":
            img = img.transpose(Image.FLIP_TOP_BOTTOM)
        elif step[0] == "affine":
            img = img.transform(img.size, Image.AFFINE, step[1], Image.NEAREST)      # F.affine, resample default
        elif step[0] == "persp":
            img = img.transform(img.size, Image.PERSPECTIVE, step[1], Image.NEAREST)
    return np.array(img)


@pytest.mark.parametrize("backend", BACKENDS)
def test_affine_gather_is_bit_identical_to_pillow(backend):
    ctx = _ctx(backend)
    rng = random.Random(5)
    B, C, H, W = (4, 19, 64, 48) if backend == "gpu" else (2, 5, 24, 20)
    lab = torch.randint(0, C, (B, H // 4, W // 4), generator=torch.Generator().manual_seed(1)).repeat_interleave(4, 1).repeat_interleave(4, 2)
    batch = O.labels_to_onehot(lab, C) if C == 19 else torch.nn.functional.one_hot(lab, C).movedim(-1, 1).float()
    maps = np.zeros((B, C, 4, 9))
    desc = {}
    for b in range(B):
        for c in range(C):
            chain, d = [], []
            for k in rng.sample(["hflip", "vflip", "affine", "persp"], 4):
                if k == "hflip" and rng.random() < 0.7:
                    chain.append(A.hflip_map(W, H)); d.append(("hflip",))
                elif k == "vflip" and rng.random() < 0.7:
                    chain.append(A.vflip_map(W, H)); d.append(("vflip",))
                elif k == "affine":
                    ang, tr, sc, sh = A.random_affine_params(W, H, rng=rng)
                    m = A.inverse_affine_matrix((W * 0.5 + 0.5, H * 0.5 + 0.5), ang, tr, sc, sh)
                    chain.append(A.affine_map(m)); d.append(("affine", m))
                elif k == "persp" and rng.random() < 0.5:
                    sp, ep = A.random_perspective_points(W, H, rng=rng)
                    co = A.perspective_coeffs(sp, ep)
                    chain.append([A.KIND_PERSPECTIVE] + co); d.append(("persp", co))
                else:
                    chain.append(A.IDENTITY)
            maps[b, c] = np.array(chain)
            desc[(b, c)] = d
    out = A.apply_maps(ctx, batch, maps).cpu().numpy()
    for (b, c), d in desc.items():
        ref = pil_chain(batch[b, c].numpy(), d)
        assert np.array_equal(out[b, c], ref), (b, c, d, int((out[b, c] != ref).sum()))
    assert set(np.unique(out)) <= {0.0, 1.0}                          # a one-hot channel stays binary


def test_parameter_draws_follow_torchvision_order():
    """RandomOrder shuffles, then each transform draws: flips one random(), affine uniform x5 (angle, dx, dy, scale,
    shear), perspective random() then 8 randints -- the call sequence of torchvision 0.4.0 on Python's `random`."""
    calls, depth = [], [0]

    def spy(name):
        def wrap(self, *a):
            if depth[0] == 0:
                calls.append(name)                 # only the transform's own calls, not Random's internal ones
            depth[0] += 1
            try:
                return getattr(random.Random, name)(self, *a)
            finally:
                depth[0] -= 1
        return wrap

    Spy = type("Spy", (random.Random,), {n: spy(n) for n in ("shuffle", "random", "uniform", "randint")})
    t = A.GpuPerChannelTransform(("hflip", "vflip", "affine", "perspective"), rng=Spy(3))
    chain = t.draw_chain(64, 64)
    assert len(chain) == 4 and calls[0] == "shuffle"
    assert calls.count("uniform") == 5 and calls.count("random") == 3 and calls.count("randint") in (0, 8)
    assert A.GpuPerChannelTransform("none").transforms == []
    assert A.GpuPerChannelTransform("all").transforms == ["vflip", "hflip", "affine", "perspective"]


@pytest.mark.parametrize("backend", BACKENDS)
def test_transform_object_runs_on_a_batch(backend):
    ctx = _ctx(backend)
    t = A.GpuPerChannelTransform(ctx=ctx, rng=random.Random(0))
    _, inputs, _ = O.synth_warp_batch(2, 32, 32, seed=3)
    out = t(inputs)
    assert out.shape == inputs.shape and set(torch.unique(out.cpu()).tolist()) <= {0.0, 1.0}
    assert not torch.equal(out.cpu(), inputs)
