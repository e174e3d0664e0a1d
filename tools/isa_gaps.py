#!/usr/bin/env python3
"""ISA shape of one kernel from a hipcc -S listing: how the vector-ALU instructions sit between the MFMAs.
  python tools/isa_gaps.py <file.s> <kernel-name-substring>
Prints instruction counts by class and the histogram of 'VALU instructions between two consecutive v_mfma'."""
import collections, re, sys
s = open(sys.argv[1]).read()
want = sys.argv[2]
for p in re.split(r'\n(?=_Z\w+:\s)', s):
    name = p.split(':')[0]
    if want not in name or not name.startswith('_Z'):
        continue
    body = p.split('.Lfunc_end')[0]
    cls = collections.Counter()
    gaps, run, seen = [], 0, False
    for ln in body.splitlines():
        t = ln.strip().split()
        if not t or t[0].startswith((';', '.', '//')) or t[0].endswith(':'):
            continue
        op = t[0]
        if op.startswith('v_mfma'):
            cls['mfma'] += 1
            if seen: gaps.append(run)
            run, seen = 0, True
        elif op.startswith('v_'):
            cls['valu'] += 1; run += 1
        elif op.startswith('ds_'):
            cls['ds'] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
            cls['vmem'] += 1
            if op.startswith('scratch_'): cls['scratch'] += 1
        elif op.startswith('s_waitcnt'):
            cls['waitcnt'] += 1
        elif op.startswith('s_barrier'):
            cls['barrier'] += 1
        elif op.startswith('s_nop'):
            cls['s_nop'] += 1
        elif op.startswith('s_'):
            cls['salu'] += 1
    print(name[:60], dict(cls))
    h = collections.Counter(min(g, 40) // 4 * 4 for g in gaps)
    tot = sum(gaps)
    print('  VALU between consecutive MFMAs (bucket: count):', ' '.join(f'{k}-{k+3}:{h[k]}' for k in sorted(h)))
    # issue-time model per wave: an MFMA gap costs max(32, 8 + 4 * valu) cycles (tools/micro/mfma_valu_overlap2)
    model = sum(max(32, 8 + 4 * g) for g in gaps)
    print(f'  gaps {len(gaps)}  VALU in gaps {tot}  issue model: sum max(32, 8 + 4 v) = {model} cycles per wave; pure MFMA {32 * len(gaps)}; MFMA + VALU sum {32 * len(gaps) + 4 * tot}')
