"""Builds swapnet_amd/csrc/libswapnet_hip.so for gfx950 (in-tree, so the .so travels with a
repo snapshot to the GPU box).  `python -m swapnet_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libswapnet_hip.so")
HIP_SOURCES = ["device.hip", "conv_gemm.hip", "wino.hip", "norm_act.hip", "losses.hip", "optim.hip", "gather.hip"]
CPP_SOURCES = ["engine.cpp", "nets.cpp", "texture.cpp", "pipeline.cpp", "gp.cpp", "capi.cpp"]
HEADERS = ["common.h", "ops.h", "hip_util.h", "engine.h", os.path.join("..", "..", "include", "swapnet_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in HIP_SOURCES + CPP_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or _newer(obj, [sp] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", sp, "-o", obj]
            if src.endswith(".cpp"):
                cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    if force or _newer(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


def build_native_driver(verbose=True):
    """tools/native_ab.cpp -> tools/_bin/native_ab: the torch-free C++ driver of the C-ABI (no Python in the process: a C2 model is
    up a second after exec, which is what short GPU calls need).  Links the in-tree library by relative rpath, so the binary
    travels with a snapshot like the .so does."""
    repo = os.path.dirname(HERE)
    src = os.path.join(repo, "tools", "native_ab.cpp")
    out = os.path.join(repo, "tools", "_bin", "native_ab")
    if not os.path.exists(src):
        return None
    if _newer(out, [src, OUT, os.path.join(repo, "include", "swapnet_hip.h")]):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", src, "-I" + os.path.join(repo, "include"), "-L" + CSRC,
               "-lswapnet_hip", "-ldl", "-Wl,-rpath,$ORIGIN/../../swapnet_amd/csrc", "-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_native_driver())
