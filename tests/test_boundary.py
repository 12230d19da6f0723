"""C-ABI boundary hygiene (CPU-only checks)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(REPO, "include", "swapnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(swn_[A-Za-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_match_binding():
    from swapnet_amd import _C
    decl = declared_symbols()
    assert len(decl) >= 40
    assert sorted(_C.Lib.exported_symbols()) == decl


def test_hip_library_builds_loads_and_exports_every_declared_symbol():
    """hipcc cross-compiles gfx950 without a GPU; dlopen works without one too (no compute call)."""
    from swapnet_amd import _C, build
    path = build.build(force=False, verbose=False)
    dll = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(dll, s), s
    lib = _C.Lib(path)
    assert lib.is_device and lib.dll.swn_abi_version() == 6
    # gfx950 code object is embedded
    out = subprocess.run(["strings", "-a", path], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_product_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from swapnet_amd import _C, engine
    with pytest.raises(_C.SwapnetHipError):
        engine.Context()                      # product library, no HIP device -> error, never a CPU path


def test_product_never_imports_the_oracle_or_the_simulator():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "swapnet_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f), errors="ignore").read()
                pats = (r"^\s*(from|import)\s+oracle", r"libswapnet_hostsim") if f.endswith(".py") else \
                    (r"#\s*include[^\n]*hostsim", r"#\s*include[^\n]*oracle")
                for pat in pats:
                    if re.search(pat, src, flags=re.M):
                        bad.append((f, pat))
    assert not bad, bad
    # importing the whole package does not pull the oracle in
    code = ("import sys; sys.path.insert(0, %r); import swapnet_amd, swapnet_amd.models, swapnet_amd.modules.swapnet_modules, "
            "swapnet_amd.optimizers, swapnet_amd.parallel, swapnet_amd.synthetic; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)" % REPO)
    subprocess.check_call([sys.executable, "-c", code])


def test_synthetic_generators_follow_the_survey_spec():
    import torch
    from oracle import swapnet_oracle as O
    from swapnet_amd import synthetic
    b = synthetic.warp_batch(2, 64, 64, seed=1234)
    ob = O.synth_warp_batch(2, 64, 64, seed=1234)
    assert torch.equal(b["bodys"], ob[0]) and torch.equal(b["input_cloths"], ob[1]) and torch.equal(b["target_cloths"], ob[2])
    assert b["target_cloths"][:, 0].abs().sum() == 0            # background = all-zero vector (data_utils.py:330-343)
    t = synthetic.texture_batch(2, 64, 64, seed=4321)
    ot = O.synth_texture_batch(2, 64, 64, seed=4321)
    assert torch.equal(t["rois"], ot[1]) and torch.equal(t["input_textures"], ot[0])
    assert (t["rois"][:, 0] == torch.tensor([63., 0., 63., 0.])).all()      # one degenerate box per sample


def test_shape_and_argument_errors_are_reported_not_crashed():
    """Error conventions at the C-ABI: bad shapes / unknown slots come back as ValueError with
    the library's message (SURVEY.md 8(b) "Error conventions"), never as a crash."""
    import torch
    from swapnet_amd import engine
    from tests import backends
    ctx = backends.hostsim_ctx()
    with pytest.raises(ValueError, match="multiples of 64"):
        engine.NativeModel(ctx, "warp", 1, 48, 48, is_train=False)          # cloth_down6 needs H/64 >= 1
    with pytest.raises(ValueError, match="power of two"):
        engine.NativeModel(ctx, "texture", 1, 96, 96, is_train=False)       # U-Net depth = log2(size)
    m = backends.get_model(ctx, "warp", 2, 64, is_train=False)
    with pytest.raises(ValueError, match="shape mismatch"):
        m.set_input(0, torch.zeros(1, 3, 64, 64))                           # batch differs from the model's
    with pytest.raises(ValueError, match="3 channels"):
        m.set_input(0, torch.zeros(2, 4, 64, 64))
    with pytest.raises(ValueError, match="unknown slot"):
        m.set_input(7, torch.zeros(2, 3, 64, 64))
    with pytest.raises(ValueError, match="training"):
        m.backward_D(0.9, 0.9)                                              # inference-only model
    with pytest.raises(ValueError, match="unknown parameter"):
        m.set_param(engine.NET_G, "no.such.weight", torch.zeros(1))
    with pytest.raises(ValueError, match="no such network"):
        m.param_infos(engine.NET_D)                                         # D exists only when is_train


def test_the_c_abi_drives_a_training_step_from_plain_cpp(tmp_path):
    """include/swapnet_hip.h from a host that is not Python: tools/native_ab.cpp (no torch, no Python in the process) compiled against
    the CI-only host-simulator build of the same C-ABI, one warp-stage G+D training step at 64 x 64 -- model creation, parameter
    enumeration / loading by state-dict name, label-map inputs, swn_model_step, the loss read-back.  (On the MI355X the same file
    links libswapnet_hip.so: profiles/native_ab_r04.txt.)"""
    from tests import backends
    so = backends.build_hostsim()
    exe = os.path.join(str(tmp_path), "native_ab_sim")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DHOSTSIM", os.path.join(REPO, "tools", "native_ab.cpp"), "-I" + os.path.join(REPO, "include"),
                           "-L" + os.path.dirname(so), "-lswapnet_hostsim", "-ldl", "-fopenmp", "-Wl,-rpath," + os.path.dirname(so), "-o", exe])
    out = subprocess.run([exe, "1", "64", "1", "0", "bench"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("bench losses")][-1]
    assert "finite 1" in line, line
    g_ce = float(line.split("G_ce")[1].split()[0])
    assert 150.0 < g_ce < 450.0, line              # 100 x the cross entropy of 19 near-uniform classes (ln 19 = 2.94) after a few steps


@pytest.mark.gpu
def test_a_context_that_owns_its_stream_packs_and_computes_what_the_callers_stream_does():
    """swn_ctx_create(create_stream = 1) -- the form a host without torch uses (tools/native_ab.cpp) -- runs on
    hipStreamNonBlocking streams, which do not order behind the null stream the allocator's zero-fill runs on.  Round 6 found the
    conditional-input channel map of PatchGAN's first layer (uploaded right behind its allocation inside swn_warp_model_create)
    zeroed by a fill that landed after the upload at the benchmark's size (24.6 GB of fills in flight): model.0.weight lost ten
    input channels and every C++-driven run computed with another discriminator (profiles/alloc_fill_race_r06.txt).  The python
    path was never affected (torch's current stream is the null stream here).  Held at C2's size, where the race was deterministic:
    both kinds of context hand back the weights they were given and compute the same step, bit for bit."""
    import torch
    from oracle import swapnet_oracle as O              # seeded weights and batch only (the checker's generators)
    from swapnet_amd import engine
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    B, H = 32, 256
    torch.manual_seed(0)
    G, D = O.warp_module_params(), O.patchgan_params(22)
    batch = O.synth_warp_batch(B, H, H, seed=4321)
    results = []
    for own in (True, False):
        ctx = engine.Context(workspace_mb=1024, use_torch_stream=not own)
        m = engine.NativeModel(ctx, "warp", B, H, H, is_train=True)
        m.load_state_dict(engine.NET_G, G)
        m.load_state_dict(engine.NET_D, D)
        back = m.state_dict(engine.NET_D, to_cpu=True)
        for k, v in D.items():
            assert torch.equal(back[k], v), ("own stream" if own else "caller's stream", k)
        m.set_hyper()
        for i, t in enumerate(batch):
            m.set_input(i, t)
        m.step((0.9, 0.8, 1.0), training=False, seed=0)
        ctx.sync()
        results.append((m.losses(), m.arena(engine.NET_D, engine.W_WEIGHT).clone(), m.output().clone()))
        m.close(); ctx.close()
    (la, wa, oa), (lb, wb, ob) = results
    assert la == lb, (la, lb)
    assert torch.equal(wa, wb) and torch.equal(oa, ob)
