"""TEST INFRASTRUCTURE ONLY.  Summary format of tests/golden/*.npz: each tensor is
stored as key/norm (L2, float64), key/sum and key/samples (24 elements at fixed
flat indices derived from an FNV hash of the key)."""
import numpy as np
import torch

NSAMP = 24


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
