"""GPU per-channel augmentation of the warp dataloader.

The reference augments the input cloth segmentation on the host, one PIL call chain per channel:
`WarpDataset._perform_cloth_transform` -> `per_channel_transform(tensor, RandomOrder([RandomVerticalFlip,
RandomHorizontalFlip, RandomAffine(degrees=10, translate=(.1,.1), scale=(.8,1.2), shear=20), RandomPerspective]))`
(/root/reference/datasets/warp_dataset.py:131-137, datasets/__init__.py:87-110, datasets/data_utils.py:346-361):
19 channels x up to 4 PIL resamplings per sample, every training step.  Here the random PARAMETERS are drawn on the
host exactly like torchvision 0.4.0's transforms draw them (same `random` calls in the same order), turned into
inverse pixel maps, and ONE device kernel (swn_op_affine_gather) applies every chain of the whole batch.

Exactness: flips and RandomAffine use PIL's NEAREST rule in Pillow's own 16.16 fixed point -> bit-identical to the
reference's per-channel PIL calls given the same parameters (tests/test_augment.py compares against Pillow itself).
RandomPerspective is resampled with NEAREST here; torchvision 0.4.0 resamples it BICUBIC and fits its coefficients
with a float32 least-squares solve, neither of which is reproducible bit for bit -- this is the one documented
deviation (a one-hot channel stays binary here, which is what the segmentation means).
"""
import math
import random

import numpy as np
import torch

from .. import _C, engine

KIND_IDENTITY, KIND_AFFINE, KIND_PERSPECTIVE = 0, 1, 2


def _fix(v):
    return float(math.floor(v * 65536.0 + 0.5))


def affine_map(coeffs):
    """Pillow Geometry.c affine_fixed: the six inverse-affine coefficients (a, b, c, d, e, f of Image.transform(AFFINE))
    -> [kind, a0, a1, a2', a3, a4, a5'] in 16.16 fixed point with the half-pixel centre folded in."""
    a = coeffs
    return [KIND_AFFINE, _fix(a[0]), _fix(a[1]), _fix(a[2] + a[0] * 0.5 + a[1] * 0.5),
            _fix(a[3]), _fix(a[4]), _fix(a[5] + a[3] * 0.5 + a[4] * 0.5), 0.0, 0.0]


def hflip_map(W, H):
    return affine_map([-1.0, 0.0, float(W), 0.0, 1.0, 0.0])        # xin = W - 1 - x (Image.transpose(FLIP_LEFT_RIGHT))


def vflip_map(W, H):
    return affine_map([1.0, 0.0, 0.0, 0.0, -1.0, float(H)])


IDENTITY = [KIND_IDENTITY] + [0.0] * 8


def inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision 0.4.0 transforms.functional._get_inverse_affine_matrix (single shear angle)."""
    angle, shear = math.radians(angle), math.radians(shear)
    scale = 1.0 / scale
    d = math.cos(angle + shear) * math.cos(angle) + math.sin(angle + shear) * math.sin(angle)
    m = [math.cos(angle + shear), math.sin(angle + shear), 0, -math.sin(angle), math.cos(angle), 0]
    m = [scale / d * v for v in m]
    m[2] += m[0] * (-center[0] - translate[0]) + m[1] * (-center[1] - translate[1])
    m[5] += m[3] * (-center[0] - translate[0]) + m[4] * (-center[1] - translate[1])
    m[2] += center[0]
    m[5] += center[1]
    return m


def random_affine_params(W, H, degrees=10, translate=(0.1, 0.1), scale=(0.8, 1.2), shear=20, rng=random):
    """RandomAffine.get_params of torchvision 0.4.0: draw order angle, dx, dy, scale, shear."""
    angle = rng.uniform(-degrees, degrees)
    max_dx, max_dy = translate[0] * W, translate[1] * H
    translations = (np.round(rng.uniform(-max_dx, max_dx)), np.round(rng.uniform(-max_dy, max_dy)))
    sc = rng.uniform(scale[0], scale[1])
    sh = rng.uniform(-shear, shear)
    return angle, translations, sc, sh


def perspective_coeffs(startpoints, endpoints):
    """Coefficients of Image.transform(PERSPECTIVE) mapping endpoints back to startpoints (torchvision
    _get_perspective_coeffs, solved in float64)."""
    A = []
    for p1, p2 in zip(endpoints, startpoints):
        A.append([p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]])
        A.append([0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]])
    B = np.array(startpoints, dtype=np.float64).reshape(8)
    return np.linalg.lstsq(np.array(A, dtype=np.float64), B, rcond=None)[0].tolist()


def random_perspective_points(W, H, distortion_scale=0.5, rng=random):
    """RandomPerspective.get_params of torchvision 0.4.0 (eight randint draws in this order)."""
    half_h, half_w = int(H / 2), int(W / 2)
    topleft = (rng.randint(0, int(distortion_scale * half_w)), rng.randint(0, int(distortion_scale * half_h)))
    topright = (rng.randint(W - int(distortion_scale * half_w) - 1, W - 1), rng.randint(0, int(distortion_scale * half_h)))
    botright = (rng.randint(W - int(distortion_scale * half_w) - 1, W - 1), rng.randint(H - int(distortion_scale * half_h) - 1, H - 1))
    botleft = (rng.randint(0, int(distortion_scale * half_w)), rng.randint(H - int(distortion_scale * half_h) - 1, H - 1))
    startpoints = [(0, 0), (W - 1, 0), (W - 1, H - 1), (0, H - 1)]
    return startpoints, [topleft, topright, botright, botleft]


class GpuPerChannelTransform:
    """Drop-in for `per_channel_transform(tensor, get_transforms(opt))` on a whole BATCH that already sits on the device.
    `input_transforms` uses the reference's option values ("hflip", "vflip", "affine", "perspective", "all", "none")."""

    def __init__(self, input_transforms=("hflip", "vflip", "affine", "perspective"), ctx=None, rng=random):
        t = [input_transforms] if isinstance(input_transforms, str) else list(input_transforms)
        every = "all" in t
        # datasets/__init__.py:get_transforms appends in this order
        self.transforms = [k for k in ("vflip", "hflip", "affine", "perspective") if every or k in t]
        if "none" in t:
            self.transforms = []
        self.ctx = ctx
        self.rng = rng

    def draw_chain(self, W, H):
        """One RandomOrder(...) call: shuffled order, then each transform's own draws.  Always returns
        len(self.transforms) maps (identity where a coin flip said no)."""
        rng = self.rng
        order = list(range(len(self.transforms)))
        rng.shuffle(order)                                          # transforms.RandomOrder.__call__
        chain = []
        for i in order:
            kind = self.transforms[i]
            if kind == "vflip":
                chain.append(vflip_map(W, H) if rng.random() < 0.5 else IDENTITY)
            elif kind == "hflip":
                chain.append(hflip_map(W, H) if rng.random() < 0.5 else IDENTITY)
            elif kind == "affine":
                angle, tr, sc, sh = random_affine_params(W, H, rng=rng)
                center = (W * 0.5 + 0.5, H * 0.5 + 0.5)
                chain.append(affine_map(inverse_affine_matrix(center, angle, tr, sc, sh)))
            else:
                if rng.random() < 0.5:
                    sp, ep = random_perspective_points(W, H, rng=rng)
                    chain.append([KIND_PERSPECTIVE] + perspective_coeffs(sp, ep))
                else:
                    chain.append(IDENTITY)
        return chain

    def draw_maps(self, B, C, W, H):
        return np.array([[self.draw_chain(W, H) for _ in range(C)] for _ in range(B)], dtype=np.float64)

    def __call__(self, batch, maps=None):
        """batch (B, C, H, W) float one-hot cloth tensor (host or device) -> transformed tensor on the device."""
        if not self.transforms:
            return batch
        ctx = self.ctx or engine.default_context()
        B, C, H, W = batch.shape
        if maps is None:
            maps = self.draw_maps(B, C, W, H)
        return apply_maps(ctx, batch, maps)


def apply_maps(ctx, batch, maps):
    """maps: array (B, C, nmaps, 9) of {kind, c0..c7} rows; see include/swapnet_hip.h swn_op_affine_gather."""
    B, C, H, W = batch.shape
    maps = np.ascontiguousarray(maps, dtype=np.float64).reshape(B * C, -1, 9)
    src = batch.detach().to(device=ctx.device, dtype=torch.float32).contiguous()
    dst = torch.empty_like(src)
    md = torch.from_numpy(maps).to(ctx.device)
    ctx.lib.call("swn_op_affine_gather", ctx.handle, _C.ptr(src), _C.ptr(dst), B, C, H, W, _C.ptr(md), maps.shape[1])
    ctx.sync()
    return dst
