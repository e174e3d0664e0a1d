#!/usr/bin/env python3
"""Raw two-queue kernel sequence around the largest main-queue gap of the analysed window (see trace_timeline.py)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void crfp::", "")[:34], r.get("Queue_Id", "0")))
mainq = max(byq, key=lambda q: sum("dcn_g8" in k[2] for k in byq[q]))
main = sorted(byq[mainq])
t0 = main[len(main) // 2][0]
win = [k for k in main if t0 <= k[0] < t0 + 15e6]
gi = max(range(len(win) - 1), key=lambda i: win[i + 1][0] - win[i][1])
gs, ge = win[gi][1], win[gi + 1][0]
allk = sorted(k for q in byq for k in byq[q] if k[1] > gs - 250e3 and k[0] < ge + 150e3)
for k in allk:
    print(f"q{k[3]} {'MAIN' if k[3]==mainq else 'side'} start {((k[0]-gs)/1e3):8.1f} us  dur {((k[1]-k[0])/1e3):6.1f} us  {k[2]}")
