"""Disassemble every gfx950 code object embedded in libcrfp_hip.so and count instructions matching a regex.
Used by tests/test_host_logic.py to assert the shipped library holds no packed-FP32 VALU ops (v_pk_fma_f32 &c.),
see crfp_amd/csrc/Makefile for why.  usage: python tools/check_isa.py [lib.so] [regex]"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """-> list of ELF byte strings (one per translation unit) for the amdgcn target."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "x.so")],
                       check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        out = []
        for i, s in enumerate(starts):
            piece = blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)]
            pf = os.path.join(td, f"b{i}.bin")
            open(pf, "wb").write(piece)
            ls = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={pf}"],
                                capture_output=True, text=True, check=True).stdout.split()
            for tgt in ls:
                if "amdgcn" not in tgt:
                    continue
                of = os.path.join(td, f"co{i}.elf")
                subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={pf}",
                                f"--targets={tgt}", f"--output={of}"], check=True)
                out.append(open(of, "rb").read())
        return out


def count(lib, pattern):
    rx = re.compile(pattern)
    n = ninstr = 0
    with tempfile.TemporaryDirectory() as td:
        for i, elf in enumerate(code_objects(lib)):
            f = os.path.join(td, f"{i}.elf")
            open(f, "wb").write(elf)
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", f], capture_output=True, text=True,
                                 check=True).stdout
            for line in dis.splitlines():
                if "\t" in line:
                    ninstr += 1
                    if rx.search(line):
                        n += 1
    return n, ninstr


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "crfp_amd", "libcrfp_hip.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else r"\bv_pk_(fma|mul|add)_f32\b"
    n, total = count(lib, pat)
    print(f"{lib}: {n} matches of /{pat}/ in {total} disassembled lines")
    sys.exit(1 if n else 0)
