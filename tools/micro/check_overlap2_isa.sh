#!/bin/bash
# Disassembly check of tools/micro/mfma_valu_overlap2.hip: per kernel, what the measured loop body holds (no s_nop, exactly 8 MFMAs
# and 8 x NF v_fma_f32 plus the 3 loop-control instructions).  Runs without a GPU.
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value --cuda-device-only -S mfma_valu_overlap2.hip -o /tmp/ov2.s 2>/dev/null
python3 - <<'P'
import re
s = open('/tmp/ov2.s').read()
for p in re.split(r'\n(?=_Z4kernILi)', s)[1:]:
    name = p.split(':')[0]
    role, nf, nt = re.match(r'_Z4kernILi(\d+)ELi(\d+)ELi(\d+)E', name).groups()
    body = p.split('.Lfunc_end')[0]
    loops = re.findall(r'(\.LBB\d+_\d+):[^\n]*Loop Header.*?s_cbranch_\w+ \1', body, re.S)
    m = re.search(r'(\.LBB\d+_\d+):[^\n]*Loop Header.*?s_cbranch_\w+ \1', body, re.S)
    loop = m.group(0) if m else body
    other = [l for l in loop.splitlines() if l.strip() and not l.strip().startswith((';', '.', 'v_mfma', 'v_fma_f32', 's_nop', '//'))]
    print(f"role {role} NF {nf:>2s} threads {nt}: loop body s_nop {len(re.findall(r's_nop', loop))}  v_mfma {len(re.findall(r'v_mfma', loop))}  "
          f"v_fma_f32 {len(re.findall(r'v_fma_f32', loop))}  other {len(other)}" + ("" if m else "  (role 3: branches per role, whole kernel counted)"))
P
