#!/usr/bin/env python3
"""Per-kernel MFMA / LDS utilisation from the rocprofv3 --pmc passes of tools/collect_util.sh.
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs); LDS busy % = SQ_LDS_IDX_ACTIVE / (kernel cycles * 256 CUs);
bank conflict % = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Kernel cycles = dispatch duration (ns) * 2.05 GHz, the shader clock
measured under MFMA load (tools/micro/mfma_rate.hip): the GRBM_GUI_ACTIVE value in the CSV is already reduced over the XCDs
(about 10x the elapsed cycles), so rocprofv3's own MfmaUtil expression cannot be re-evaluated from it."""
import csv, glob, os, re, sys, collections

out = sys.argv[1]
CUS, SIMDS = 256, 1024


def load(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    files = glob.glob(os.path.join(out, sub, "*", "*counter_collection.csv"))
    for f in files:
        seen = set()
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void crfp::", "").replace("crfp::", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), name)
            if key not in seen:
                seen.add(key); calls[name] += 1
                acc[name]["_cycles"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 2.05
    return acc, calls


mf, calls = load("mfma")
ld, _ = load("lds")
print(f"{'kernel':44s} {'calls':>6s} {'MFMA busy %':>12s} {'LDS busy %':>11s} {'bank confl %':>13s}")
rows = []
for k in mf:
    g = mf[k].get("_cycles", 0.0)
    if g <= 0:
        continue
    mfma = 100.0 * mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * SIMDS)
    g2 = ld.get(k, {}).get("_cycles", 0.0)
    lds = 100.0 * ld[k].get("SQ_LDS_IDX_ACTIVE", 0.0) / (g2 * CUS) if g2 else float("nan")
    idx = ld.get(k, {}).get("SQ_LDS_IDX_ACTIVE", 0.0)
    bc = 100.0 * ld[k].get("SQ_LDS_BANK_CONFLICT", 0.0) / idx if idx else float("nan")
    rows.append((g, k, calls[k], mfma, lds, bc))
for g, k, c, mfma, lds, bc in sorted(rows, reverse=True)[:14]:
    print(f"{k[:44]:44s} {c:6d} {mfma:12.1f} {lds:11.1f} {bc:13.1f}")

co, _ = load("coexec")
if co:
    print()
    print(f"{'kernel':44s} {'COEXEC / MFMA-busy %':>20s} {'WAIT_ANY / WAVE_CYCLES %':>25s} {'VALU active / WAVE_CYCLES %':>28s}")
    for g, k, c, mfma, lds, bc in sorted(rows, reverse=True)[:14]:
        d = co.get(k)
        if not d:
            continue
        busy = mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * co[k].get("_cycles", 1.0) / max(mf[k].get("_cycles", 1.0), 1.0)
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        print(f"{k[:44]:44s} {100.0 * d.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0.0) / busy if busy else float('nan'):20.1f} "
              f"{100.0 * d.get('SQ_WAIT_ANY', 0.0) / wc if wc else float('nan'):25.1f} {100.0 * d.get('SQ_ACTIVE_INST_VALU', 0.0) / wc if wc else float('nan'):28.1f}")
