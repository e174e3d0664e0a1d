#!/usr/bin/env python3
"""Generates the instruction-group schedule of dcn_fused_kernel (crfp_amd/csrc/gather.hip, `#include "dcn_fused_schedule.inc"`).

The kernel computes the 32 -> 216 offset / mask head one cout tile T (0..6) and 16-channel chunk CH (0, 1) at a time: 9 taps of
3 MFMAs per (T, CH) stage.  Everything else -- activating the raw sums of the previous tile (TR), the coordinates + gathers of a
sampling position (I), its bilinear blend (L), the fp16 split + DCN MFMAs of a pair (S) -- is vector-ALU work that must sit
INSIDE the MFMA gaps to be hidden (tools/micro/mfma_valu_overlap2: <= 6 vector instructions per 32-cycle gap are free, a
burst of them after a burst of MFMAs is not).  Round 2 placed whole pairs (60-140 vector instructions) between groups of 6-12
MFMAs; this list scheduler places ONE micro-item (<= ~30 vector instructions) after EVERY tap (3 MFMAs).

Dependencies (all indices compile-time):
  RAW(T)     after the last tap of stage (T, 1); it frees the accumulators for tile T + 1
  TR(T, q)   activates slots 16 T + 4 q .. + 3 (slot 3 p + c = component c of the lane half's position p); after RAW(T)
  I(u, pi)   position 2 u + pi needs slots 3 (2u + pi) .. + 2 activated; writes pair buffer Q[u & 1]: after L(u - 2, *)
  L(u, pi)   at least GAP taps after I(u, pi) (the gathers' latency); writes xs[4 pi ..]: after S(u - 1)
  S(u)       after L(u, 0), L(u, 1); in pair order (the DCN accumulation order is part of the bit-exact contract)
"""
import sys

GAP = int(sys.argv[1]) if len(sys.argv) > 1 else 5
TAIL_I_FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def build():
    taps = [(T, ch, tap) for T in range(7) for ch in range(2) for tap in range(9)]
    done_at = {}          # item -> tap index after which it ran
    sched = {i: [] for i in range(len(taps))}
    raw_at = {T: taps.index((T, 1, 8)) for T in range(7)}
    items = []
    for T in range(7):
        for q in range(4):
            if 16 * T + 4 * q < 108:
                items.append(("TR", T, q))
    for u in range(18):
        items += [("I", u, 0), ("I", u, 1), ("L", u, 0), ("L", u, 1), ("S", u)]

    def slots_ready(p, t):
        for s in range(3 * p, 3 * p + 3):
            it = ("TR", s // 16, (s % 16) // 4)
            if it not in done_at or done_at[it] > t:
                return False
        return True

    def ready(it, t):
        k = it[0]
        if k == "TR":
            return raw_at[it[1]] < t or (raw_at[it[1]] == t and False)
        if k == "I":
            _, u, pi = it
            if not slots_ready(2 * u + pi, t - 1):
                return False
            if pi == 1 and ("I", u, 0) not in done_at:
                return False
            if u >= 2 and not all(("L", u - 2, x) in done_at for x in (0, 1)):
                return False
            return True
        if k == "L":
            _, u, pi = it
            if ("I", u, pi) not in done_at or t - done_at[("I", u, pi)] < GAP:
                return False
            if pi == 1 and ("L", u, 0) not in done_at:
                return False
            if u >= 1 and ("S", u - 1) not in done_at:
                return False
            return True
        if k == "S":
            _, u = it
            return ("L", u, 0) in done_at and ("L", u, 1) in done_at and (u == 0 or ("S", u - 1) in done_at)
    prio = {"S": 0, "L": 1, "I": 2, "TR": 3}
    pending = list(items)
    tail = []
    t = 0
    while pending:
        if t < len(taps):
            T, ch, tap = taps[t]
            if T == 0:      # tile 0 has nothing to hide yet
                t += 1
                continue
        cands = [it for it in pending if ready(it, t)]
        if cands:
            # behind the last tap nothing hides a gather's latency any more: issue every position whose pair buffer is free (I) before blending
            # anything (L), so that the tail pays ONE exposed round trip instead of one per pair (round 5; TAIL_I_FIRST=0: the round-3 order)
            pr = dict(prio, I=-1, TR=-2) if (TAIL_I_FIRST and t >= len(taps)) else prio
            it = min(cands, key=lambda x: (pr[x[0]], pending.index(x)))
            pending.remove(it)
            done_at[it] = t
            (sched[t] if t < len(taps) else tail).append((t, it))
            if it[0] != "TR" and t < len(taps):     # a tap takes one sampler item plus, if one is ready, one activation item
                tr = [x for x in pending if x[0] == "TR" and ready(x, t)]
                if tr:
                    pending.remove(tr[0])
                    done_at[tr[0]] = t
                    sched[t].append((t, tr[0]))
        elif t >= len(taps):
            tail.append((t, ("WAIT",)))
        t += 1
        if t > 400:
            raise SystemExit("schedule does not converge")
    return taps, sched, tail


def emit(taps, sched, tail):
    out = []
    Q = lambda u: f"Q{u & 1}"
    def macro(it):
        k = it[0]
        if k == "TR": return f"DF_TR({it[1]}, {it[2]})"
        if k == "I": return f"DF_I1({it[1]}, {it[2]}, {Q(it[1])})"
        if k == "L": return f"DF_L({it[1]}, {it[2]}, {Q(it[1])})"
        if k == "S": return f"DF_S({it[1]})"
        return ""
    for i, (T, ch, tap) in enumerate(taps):
        line = ""
        if tap == 0:
            if ch == 0 and T > 0:
                line += f"DF_BIAS({T}) "
            line += f"DF_BEGIN({T}, {ch}) "
            if T == 0 and ch == 0:
                line += "DF_BIAS(0) "
        # one scheduling region = the item(s) followed by the tap: the tap's LDS operand reads can rise above the item's vector
        # work and its MFMAs can sink into it; the fence in front keeps items and taps in this order
        if sched[i]:
            line += "DF_SB " + " ".join(macro(it) for _, it in sched[i]) + " "
        line += f"DF_TAPS({ch}, {tap}, {tap + 1}) "
        if ch == 1 and tap == 8:
            line += f"DF_SB DF_RAW({T}) "
        out.append("    " + line.rstrip())
    # the tail (behind the last tap).  Round 6: two hooks for the persistent form of the kernel (no-ops otherwise) -- DF_TAIL_REQUEST behind the
    # last gather issue (the next tile's halo tile and weight stage 0 are requested), DF_TAIL_COMMIT behind the last DCN MFMA (they go to LDS)
    line = "    "
    real = [it for _, it in tail if it[0] != "WAIT"]
    last_i = max(i for i, it in enumerate(real) if it[0] == "I")
    for i, it in enumerate(real):
        line += macro(it) + " DF_SB "
        if i == last_i:
            line += "DF_TAIL_REQUEST "
    line += "DF_TAIL_COMMIT"
    out.append(line.rstrip())
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    taps, sched, tail = build()
    used = sum(1 for v in sched.values() if v)
    sys.stderr.write(f"GAP {GAP}: {used} of {len(taps) - 18} taps carry an item; tail items {sum(1 for _, it in tail if it[0] != 'WAIT')}: "
                     f"{[it for _, it in tail if it[0] != 'WAIT']}\n")
    sys.stdout.write("// generated by tools/gen/dcn_fused_schedule.py (GAP = %d taps between a position's gathers and its blend) -- do not edit\n" % GAP)
    sys.stdout.write(emit(taps, sched, tail))
