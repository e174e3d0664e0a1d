#!/usr/bin/env python3
"""Largest idle gaps on the busiest queue of a rocprofv3 --kernel-trace CSV, with the kernels on either side.
usage: trace_main_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 8e3
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]))
main = max(byq.values(), key=len)
main.sort()
main = main[len(main) // 2:]     # steady-state clips
span = main[-1][1] - main[0][0]
busy = sum(e - s for s, e, _ in main)
print(f"main queue: {len(main)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms")
agg = collections.Counter(); cnt = collections.Counter()
for i in range(len(main) - 1):
    g = main[i + 1][0] - main[i][1]
    if g > min_gap:
        key = (main[i][2], main[i + 1][2])
        agg[key] += g; cnt[key] += 1
for (a, b), g in agg.most_common(12):
    print(f"  {g/1e3:9.1f} us in {cnt[(a,b)]:3d} gaps   after {a}   before {b}")
small = sum(main[i + 1][0] - main[i][1] for i in range(len(main) - 1) if 0 < main[i + 1][0] - main[i][1] <= min_gap)
print(f"  gaps <= {min_gap/1e3:.0f} us: {small/1e6:.3f} ms total")
