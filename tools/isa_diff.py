"""Per-kernel ISA comparison of two builds of one HIP translation unit (round 4, written when device code had to be added without a
GPU to run it on: every kernel the default routing launches must come out of the compiler instruction for instruction as before).

  python tools/isa_diff.py before.o after.o        # objects as hipcc -c leaves them (swapnet_amd/csrc/build/*.hip.o)

Extracts the gfx950 code object from each object's .hip_fatbin section, disassembles it and compares every function by its
instruction text (addresses and encodings dropped).  Prints the functions that differ, disappear or are new; exit status 1 if any
existing function changed."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def disassemble(obj, tmp, tag):
    fat = os.path.join(tmp, tag + ".fat")
    co = os.path.join(tmp, tag + ".co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co, "--unbundle"])
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is not None:
            t = re.sub(r"//.*$", "", line).strip()
            # the literal of the s_add_u32 / s_addc_u32 pair behind s_getpc_b64 is the distance to a constant table or another
            # function: it moves when a kernel is added to the translation unit, the instruction does not change
            if cur is not None and funcs[cur] and re.match(r"s_addc?_u32 s\d+, s\d+, 0x[0-9a-f]+$", t) and \
                    any(p.startswith("s_getpc_b64") for p in funcs[cur][-3:]):
                t = re.sub(r"0x[0-9a-f]+$", "<pc-relative>", t)
            if t:
                funcs[cur].append(t)
    return funcs


def main(before, after):
    with tempfile.TemporaryDirectory() as tmp:
        a, b = disassemble(before, tmp, "a"), disassemble(after, tmp, "b")
    changed = [k for k in a if k in b and a[k] != b[k]]
    gone = [k for k in a if k not in b]
    new = [k for k in b if k not in a]
    def demangle(names):
        tool = os.path.join(LLVM, "llvm-cxxfilt")
        if not names or not os.path.exists(tool):
            tool = "c++filt" if names else None
        if not tool:
            return []
        try:
            return subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        except OSError:
            return names
    print("unchanged %d   changed %d   removed %d   new %d" % (len(a) - len(changed) - len(gone), len(changed), len(gone), len(new)))
    for tag, names in (("CHANGED", changed), ("REMOVED", gone), ("NEW", new)):
        for n in demangle(names):
            if n:
                print(" ", tag, n[:160])
    return 1 if changed or gone else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
