"""TEST INFRASTRUCTURE ONLY.  Summary format of tests/golden/*.npz: each tensor is
stored as key/norm (L2, float64), key/sum and key/samples (24 elements at fixed
flat indices derived from an FNV hash of the key).  A few tensors are ALSO stored
whole as key/full (float32; FULL_TENSORS below) so that a localized error -- one tile,
one channel block -- that leaves the norm and 24 samples of a large tensor intact
cannot hide: the generator output, the gradients at both ends of the backward chain
(tail conv, first encoder convs) and the discriminator head."""
import numpy as np
import torch

NSAMP = 24
FULL_TENSORS = {
    "warp": ("step0/fakes", "step1/fakes", "step0/gradG/upsample_and_pad.2.weight", "step0/postG/upsample_and_pad.2.weight",
             "step0/gradG/body_down1.model.0.weight", "step0/gradG/cloth_down1.model.0.weight",
             "step0/gradD/model.11.weight", "step0/gradD/model.0.weight"),
    "texture": ("step0/fakes", "step0/gradG/encode.model.0.weight", "step0/gradG/unet.model.model.3.weight",
                "step0/postG/unet.model.model.3.weight", "step0/gradD/model.11.weight", "step0/gradD/model.0.weight"),
}


# The fixtures recorded at the benchmarked resolution (256 x 256: BASELINE.json C2 / C3) and at C1's batch (64 x 64, bs 4).  Same
# choice of tensors; the generator output is stored at stride 4 in the 256 x 256 files (store_full).
FULL_TENSORS["warp_256"] = tuple(k for k in FULL_TENSORS["warp"] if k.startswith("step0/"))
FULL_TENSORS["warp_c1"] = ("step0/fakes",)
FULL_TENSORS["texture_256"] = FULL_TENSORS["texture"]
FULL_STRIDE = {"warp_256": 4, "texture_256": 4}


def _fnv(s):
    h = 2166136261
    for c in s.encode():
        h = ((h ^ c) * 16777619) & 0xFFFFFFFF
    return h


def sample_idx(numel, key):
    rs = np.random.RandomState(_fnv(key) % (2 ** 31))
    return rs.randint(0, numel, size=NSAMP)


def summarize(out, key, t):
    t = t.detach().double().cpu().reshape(-1)
    idx = sample_idx(t.numel(), key)
    out[key + "/norm"] = np.float64(t.norm().item())
    out[key + "/sum"] = np.float64(t.sum().item())
    out[key + "/samples"] = t[torch.from_numpy(idx)].numpy()


def store_full(out, key, t, stride=1):
    """stride > 1 (the 256 x 256 fixtures): every stride-th row and column of the two trailing axes is kept -- a 2 x 19 x 256 x 256
    generator output is 10 MB whole, 0.6 MB at stride 4 -- and key/stride records it; compare_full subsamples the same way."""
    t = t.detach().float().cpu()
    if stride > 1:
        t = t[..., ::stride, ::stride].contiguous()
        out[key + "/stride"] = np.int64(stride)
    out[key + "/full"] = t.numpy()


def compare_full(gold, key, t, rtol=1e-3, flip_slices=0, flip_cap=1e-2, outlier_frac=0.0):
    """Whole-tensor check against key/full: rel-L2 <= rtol AND every element within
    rtol*|ref| + 4*rtol*rms(ref) (round-off outliers of a few sigma pass, a wrong tile / channel -- an error
    of order rms -- cannot).  flip_slices > 0 (gradient tensors): if that fails, the same two criteria are applied
    with the `flip_slices` output-channel slices (dim 0) of largest squared error left out, and the whole tensor must
    stay within flip_cap -- the signature of single LeakyReLU / ReLU sign flips between two fp32 evaluations
    (tests/backends.py assert_grads_vs_fp64), not of a diffuse error.  outlier_frac > 0 (un-pinned gradients at 256 x 256 only): that
    fraction of the elements may sit outside the per-element bar -- at the far end of the backward chain every upstream flip arrives
    spread over all channels, a heavy-tailed error whose rel-L2 stays inside rtol -- and the message counts them.
    Returns (ok, message)."""
    ref = torch.from_numpy(np.asarray(gold[key + "/full"])).double()
    t = t.detach().double().cpu()
    if key + "/stride" in gold:
        st = int(gold[key + "/stride"])
        t = t[..., ::st, ::st]
    if tuple(t.shape) != tuple(ref.shape):
        return False, "%s: shape %s vs reference %s" % (key, tuple(t.shape), tuple(ref.shape))
    rms = float(ref.norm()) / max(np.sqrt(ref.numel()), 1.0)
    err = (t - ref).abs()
    tol = rtol * ref.abs() + 4 * rtol * rms
    rl2 = float((t - ref).norm() / (ref.norm() + 1e-30))
    bad = int((err > tol).sum())
    ok = rl2 <= rtol and bad <= outlier_frac * ref.numel()
    msg = "%s: rel-L2 %.2e (tol %.0e), %d / %d elements outside tol, worst |d| %.3e" % (
        key, rl2, rtol, bad, ref.numel(), float(err.max()))
    if not ok and flip_slices > 0 and t.dim() >= 1 and t.shape[0] > flip_slices and rl2 <= flip_cap:
        d2 = (err ** 2).reshape(t.shape[0], -1).sum(dim=1)
        keep = d2.argsort()[: t.shape[0] - flip_slices]
        rl2k = float(d2[keep].sum().sqrt() / (ref.norm() + 1e-30))
        badk = int((err[keep] > tol[keep]).sum())
        ok = rl2k <= rtol and badk <= outlier_frac * ref.numel()
        msg += "; without the %d worst output channels: rel-L2 %.2e, %d outside tol (sign-flip signature %s)" % (
            flip_slices, rl2k, badk, "accepted" if ok else "NOT met")
    return ok, msg


def compare(gold, key, t, rtol=1e-3, atol_frac=1e-3):
    """Relative check of tensor `t` (NCHW, reference layout) against a stored summary.
    norm: relative; samples: |d| <= rtol*|ref| + atol_frac*rms(ref tensor).
    Returns (ok, message)."""
    t = t.detach().double().cpu().reshape(-1)
    gn = float(gold[key + "/norm"])
    gs = np.asarray(gold[key + "/samples"], dtype=np.float64)
    idx = sample_idx(t.numel(), key)
    mine = t[torch.from_numpy(idx)].numpy()
    rms = gn / max(np.sqrt(t.numel()), 1.0)
    n = float(t.norm())
    err_n = abs(n - gn) / max(gn, 1e-30)
    err_s = np.abs(mine - gs)
    tol_s = rtol * np.abs(gs) + atol_frac * rms
    ok = (err_n <= rtol or gn < 1e-30 and n < 1e-20) and bool(np.all(err_s <= tol_s + 1e-30))
    msg = "%s: norm ref %.6e got %.6e (rel %.2e); worst sample err %.3e (tol %.3e)" % (
        key, gn, n, err_n, float(err_s.max()), float(tol_s[err_s.argmax()]))
    return ok, msg
