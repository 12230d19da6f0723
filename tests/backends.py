"""Test helpers: the product (HIP) context and the CI-only host-simulator context."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM_DIR = os.path.join(REPO, "tests", "hostsim")
HOSTSIM_SO = os.path.join(HOSTSIM_DIR, "build", "libswapnet_hostsim.so")
_CSRC = os.path.join(REPO, "swapnet_amd", "csrc")
_HOST_SOURCES = [os.path.join(_CSRC, f) for f in ("engine.cpp", "nets.cpp", "texture.cpp", "pipeline.cpp", "gp.cpp", "capi.cpp")] + \
    [os.path.join(HOSTSIM_DIR, "hostsim_ops.cpp")]


def build_hostsim():
    deps = _HOST_SOURCES + [os.path.join(_CSRC, h) for h in ("ops.h", "common.h", "engine.h")]
    if os.path.exists(HOSTSIM_SO) and all(os.path.getmtime(d) <= os.path.getmtime(HOSTSIM_SO) for d in deps):
        return HOSTSIM_SO
    os.makedirs(os.path.dirname(HOSTSIM_SO), exist_ok=True)
    # linked under a per-process name and renamed into place: concurrent builders (pytest-xdist workers, the two gloo ranks of the
    # data-parallel tests) each publish a complete file, and nobody dlopens one that is still being written ("file too short")
    tmp = "%s.%d.tmp" % (HOSTSIM_SO, os.getpid())
    try:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-o", tmp] + _HOST_SOURCES)
        os.replace(tmp, HOSTSIM_SO)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    return HOSTSIM_SO


_ctx = {}


def hostsim_ctx(fresh=False):
    from swapnet_amd import _C, engine
    if fresh:
        return engine.Context(lib=_C.Lib(build_hostsim()), workspace_mb=64)
    if "sim" not in _ctx:
        _ctx["sim"] = engine.Context(lib=_C.Lib(build_hostsim()), workspace_mb=256)
    return _ctx["sim"]


def gpu_ctx():
    import torch
    from swapnet_amd import engine
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    if "gpu" not in _ctx:
        _ctx["gpu"] = engine.Context(workspace_mb=1024)
    return _ctx["gpu"]


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rel_l2_without_worst_slices(a, b, k):
    """rel-L2(a, b) after dropping the k output-channel slices (dim 0) that carry the most squared error."""
    a, b = a.double().cpu(), b.double().cpu()
    if a.dim() == 0 or a.shape[0] <= k:
        return float("inf")          # nothing left to judge: no exemption (as oracle/golden_io.py compare_full)
    d2 = ((a - b) ** 2).reshape(a.shape[0], -1).sum(dim=1)
    keep = d2.argsort()[: a.shape[0] - k]
    return float(d2[keep].sum().sqrt() / (b.norm() + 1e-30))


FLIP_SLICES = 3        # single activation sign flips tolerated per tensor at the small sizes (see below)
FLIP_CAP = 1e-2        # ... and what they may cost the whole tensor


def assert_grads_vs_fp64(got, ref32, ref64, skip, what, floor=1e-3, mult=2.0, cap=None):
    """The UN-PINNED gradient bar: per tensor, rel-L2(native, fp64 oracle) <= max(floor, 2 x rel-L2(torch fp32 oracle,
    fp64 oracle)).  floor = 1e-3 (north_star) at the small sizes.  At 256x256 a handful of LeakyReLU / ReLU sign flips
    between ANY two fp32 evaluations (2-6 elements out of 2M on PatchGAN's 31x31 map) are the whole distance to the fp64
    gradient -- a Poisson count proportional to the forward round-off, 0.4e-3 .. 4e-3 for torch's own fp32 backward,
    1.1e-3 .. 3.9e-3 here under the default kernel routing (round 4: PatchGAN 2.0-2.6 x torch's own distance, the generator
    0.9-1.05 x) -- so those call sites pass mult = 4, cap = 5e-3: per tensor four times torch-fp32's distance, never more than
    the old flat 5e-3; the rigorous comparison is the pinned one (tests/test_pattern_replay.py: activation pattern replayed in
    the oracle, tolerance 1e-4).
    At the small sizes flips are rare but not impossible (test_pattern_replay counts 0-2 per network at 64x64): ONE flipped
    element of a layer's output changes that layer's weight / bias gradient in ONE output channel by O(1) of that channel,
    2e-3 .. 6e-3 of the tensor for PatchGAN's first 64-channel layer.  A tensor over the bar therefore still passes if the
    excess is exactly that signature: without its FLIP_SLICES worst output-channel slices it meets the bar, and as a
    whole it stays under FLIP_CAP.  Anything diffuse (a precision loss, a wrong term) fails as before.
    Returns (worst native, worst torch-fp32)."""
    w_hip = w_t32 = 0.0
    rows = []
    for k, v in ref64.items():
        if skip(k):
            continue
        e_hip, e_t32 = rel_l2(got[k], v), rel_l2(ref32[k], v)
        w_hip, w_t32 = max(w_hip, e_hip), max(w_t32, e_t32)
        rows.append((e_hip, e_t32, k))
    bad = []
    for e_hip, e_t32, k in rows:
        bar = max(floor, mult * e_t32)
        if cap is not None:
            bar = min(bar, cap)
        if e_hip <= bar:
            continue
        rest = rel_l2_without_worst_slices(got[k], ref64[k], FLIP_SLICES)
        if e_hip <= FLIP_CAP and rest <= bar:
            print("%s  %-60s native %.2e over the bar %.2e, %.2e without its %d worst output channels: sign-flip signature, accepted"
                  % (what, k, e_hip, bar, rest, FLIP_SLICES))
            continue
        bad.append((e_hip, e_t32, k))
    if bad or os.environ.get("SWAPNET_TEST_VERBOSE"):
        for e_hip, e_t32, k in sorted(rows, reverse=True)[:12]:
            print("%s  %-60s native %.2e  torch fp32 %.2e  ratio %.2f" % (what, k, e_hip, e_t32, e_hip / max(e_t32, 1e-30)))
    assert not bad, (what, [(k, "native %.2e" % a, "torch fp32 %.2e" % b) for a, b, k in bad])
    return w_hip, w_t32


_models = {}


def get_model(ctx, kind, B, H, is_train=True):
    """Native models are expensive to create on the host simulator (GBs of zeroed arenas): the
    parity tests share one per (backend, stage, shape) and reset its training state instead."""
    from swapnet_amd import engine
    key = (id(ctx), kind, B, H, is_train, os.environ.get("SWN_WINO_MINC"))      # a model keeps the routing it was built under
    if key not in _models:
        _models[key] = engine.NativeModel(ctx, kind, B, H, H, is_train=is_train)
    return _models[key]


def reset_state(m, state_dicts):
    """Load weights, zero both Adam moments and the step counters, default hyper-parameters."""
    import torch
    from swapnet_amd import engine
    for net, sd in state_dicts.items():
        m.load_state_dict(net, sd)
        if net != engine.NET_VGG and m.is_train:
            # both Adam moments := 0 on the flat arenas (one fill each instead of one packed upload per parameter; it also clears the
            # alignment pads between parameters, which no named parameter covers)
            m.arena(net, engine.W_EXP_AVG).zero_()
            m.arena(net, engine.W_EXP_AVG_SQ).zero_()
            m.ctx.sync()
            m.optim_step_count(net, 0)
    if m.is_train:
        m.set_hyper()


def collect_patterns(m, vgg=False):
    """The activation / pooling branches of the pass the model just ran, grouped the way the step oracles replay them
    (oracle.swapnet_oracle.PatternReplay): G, D (D step, batch [fake | real]), D_G (D inside the G step), VGG."""
    groups = {"G": m.act_patterns(0), "D": m.act_patterns(1), "D_G": m.act_patterns(2)}
    if vgg:
        groups["VGG"] = m.act_patterns(3)
    return groups


def assert_grads_replayed(got, ref64, skip, tol, what):
    """With the native pass's activation pattern replayed in the float64 oracle, every gradient tensor agrees to `tol`
    rel-L2 (what is left is the fp32 arithmetic of the kernels).  Returns the worst error."""
    worst = 0.0
    for k, v in ref64.items():
        if skip(k):
            continue
        e = rel_l2(got[k], v)
        worst = max(worst, e)
        assert e <= tol, (what, k, "rel-L2 %.2e > %.1e" % (e, tol))
    return worst


# ---- kernel routing of the full-size tests ------------------------------------------------------------------------------
_ROUTE_SCRIPT = r"""
import sys, json
sys.path.insert(0, %(repo)r)
import torch
from swapnet_amd import engine, synthetic
kind, B, H = %(kind)r, %(B)d, %(H)d
ctx = engine.Context(workspace_mb=1024)
m = engine.NativeModel(ctx, kind, B, H, H, is_train=True)
m.set_hyper()
synthetic.fill_inputs(m, kind, B, H, H, seed=1)
ctx.route_trace(True)
m.forward(True, 3); m.backward_D(0.9, 0.8); m.optimizer_step(1); m.backward_G(1.0); m.optimizer_step(0)
ctx.sync(); ctx.route_trace(False)
print("ROUTE" + json.dumps(ctx.route_report()))
"""
_default_routes = {}


def default_route(kind, B, H):
    """The launch list of one phased training step of a `kind` model built and run in a SEPARATE process whose environment
    carries no SWN_* variable at all -- what `python bench.py` / smoke() / a user gets.  Cached per (kind, B, H)."""
    import json
    import sys
    key = (kind, B, H)
    if key not in _default_routes:
        env = {k: v for k, v in os.environ.items() if not k.startswith("SWN_")}
        out = subprocess.run([sys.executable, "-c", _ROUTE_SCRIPT % dict(repo=REPO, kind=kind, B=B, H=H)], env=env, check=True,
                             capture_output=True, text=True, timeout=900).stdout
        line = [l for l in out.splitlines() if l.startswith("ROUTE")][-1]
        _default_routes[key] = json.loads(line[5:])
    return _default_routes[key]


class traced_route:
    """with traced_route(ctx) as r: <the step under test>; r.lines is its launch list."""

    def __init__(self, ctx):
        self.ctx, self.lines = ctx, []

    def __enter__(self):
        self.ctx.route_trace(True)
        return self

    def __exit__(self, *exc):
        self.ctx.sync()
        self.ctx.route_trace(False)
        self.lines = self.ctx.route_report()
        return False


def assert_default_routing(lines, kind, B, H):
    """The step just checked against the oracle launched exactly what a default-environment process launches for this model
    (the parity suite used to run under SWN_WINO_MINC=32, which re-routes three of C2's largest GEMMs)."""
    leaked = sorted(k for k in os.environ if k.startswith("SWN_") and k not in ("SWN_PROF_DETAIL", "SWN_PC_DEBUG"))
    assert not leaked, ("kernel-routing switches set in a default-routing test", leaked)
    want = default_route(kind, B, H)
    assert len(want) > 20, want
    missing = [l for l in want if l not in lines]
    extra = [l for l in lines if l not in want]
    assert not missing and not extra, ("launch list differs from the default-environment one", missing[:8], extra[:8])
