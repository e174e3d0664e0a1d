#!/usr/bin/env python3
"""Idle gaps between consecutive kernels per queue from a rocprofv3 --kernel-trace CSV (steady-state clips only)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in byq.items():
    ks.sort()
    if len(ks) < 200:
        continue
    ks = ks[len(ks) // 3:]          # skip warm-up / packing
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g > 0]
    pos.sort()
    small = [g for g in pos if g < 200000]
    neg = [g for g in gaps if g <= 0]
    print(f"   gaps < 200 us: n={len(small)} total {sum(small)/1e6:.3f} ms; back-to-back (start <= previous end): n={len(neg)} mean overlap {-sum(neg)/max(1,len(neg))/1e3:.2f} us")
    print(f"queue {q}: {len(ks)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), "
          f"gaps: n={len(pos)} total {sum(pos)/1e6:.2f} ms median {pos[len(pos)//2]/1e3:.2f} us p90 {pos[int(len(pos)*0.9)]/1e3:.2f} us max {pos[-1]/1e3:.1f} us")
