#!/usr/bin/env python3
"""Disassemble the gfx950 code object(s) inside a host object / shared library / executable built by hipcc and print one kernel's
ISA (or every kernel's size).  usage: python tools/kdis.py <file> [substring of the demangled kernel name]"""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin"
data = open(sys.argv[1], "rb").read()
want = sys.argv[2] if len(sys.argv) > 2 else None
idx = [m.start() for m in re.finditer(b"\x7fELF", data)] + [len(data)]
seen = set()
for i in range(len(idx) - 1):
    blob = data[idx[i]:idx[i + 1]]
    if blob[18:20] != b"\xe0\x00":      # e_machine == EM_AMDGPU (224)
        continue
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(blob)
    if want is None:
        out = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", f.name], capture_output=True, text=True).stdout
        for l in out.splitlines():
            p = l.split()
            if len(p) >= 8 and p[3] == "FUNC" and p[7] not in seen:
                seen.add(p[7])
                print(p[2], subprocess.run(["c++filt", p[7]], capture_output=True, text=True).stdout.strip()[:110])
    else:
        out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", f.name], capture_output=True, text=True).stdout
        on = False
        for l in out.splitlines():
            if l.endswith(">:"):
                on = want in l and l not in seen
                if on:
                    seen.add(l)
            if on:
                print(l)
    os.unlink(f.name)
