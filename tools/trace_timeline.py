#!/usr/bin/env python3
"""For one steady-state clip of a rocprofv3 --kernel-trace CSV: every idle gap > N us on the queue that runs the DCN
kernels (the caller's stream), with what the other queue (side stream) was doing meanwhile.
usage: trace_timeline.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void crfp::", "")[:40]))
mainq = max(byq, key=lambda q: sum("dcn_g8" in k[2] for k in byq[q]))
sideq = max((q for q in byq if q != mainq), key=lambda q: sum("hr_prep" in k[2] for k in byq[q]))
main = sorted(byq[mainq]); side = sorted(byq[sideq])
# clip boundaries: psnr / first-frame pattern is hard to see; take the last 2/3 and print 14 ms of it
t0 = main[len(main) // 2][0]
win = [k for k in main if t0 <= k[0] < t0 + 15e6]
print(f"main queue {mainq}: {len(main)} kernels; side queue {sideq}: {len(side)} kernels; window of {len(win)} kernels")
idle = 0
for i in range(len(win) - 1):
    g = win[i + 1][0] - win[i][1]
    if g > 0: idle += g
    if g > min_gap:
        gs, ge = win[i][1], win[i + 1][0]
        ov = [k for k in side if k[1] > gs and k[0] < ge]
        last_side_end = max((k[1] for k in side if k[1] <= ge), default=0)
        print(f"t={((gs-t0)/1e6):7.3f} ms gap {g/1e3:7.1f} us  after {win[i][2]:40s} before {win[i+1][2]:40s} | side: {len(ov)} kernels"
              + (f" [{ov[0][2]} .. {ov[-1][2]}], last side end {((last_side_end-gs)/1e3):.1f} us after gap start" if ov else f", idle (last side kernel ended {((gs-last_side_end)/1e3):.1f} us before the gap)"))
print(f"idle in window: {idle/1e6:.3f} ms of {(win[-1][1]-win[0][0])/1e6:.3f} ms")
