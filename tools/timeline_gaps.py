#!/usr/bin/env python3
"""Steady-state timeline of the TSDF batches from a rocprofv3 kernel trace (csv): per update launch its duration, the gap to the
previous update launch, and which prep kernels ran inside that gap; then the per-batch critical path.
    tools/timeline_gaps.py <dir-with-*kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
tr = []
for r in csv.DictReader(open(f)):
    if "tl3d" in r["Kernel_Name"]:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tl3d::", "").split("<")[0]
        tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, r.get("Queue_Id", "?")))
tr.sort()
upd = [x for x in tr if x[2] == "tsdf_update_pairs_kernel"]
print("update launches", len(upd))
sel = upd[len(upd) // 3: len(upd) // 3 + 24]
t0 = sel[0][0]
prev_end = None
for s, e, nm, q in sel:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    line = f"update start {(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap-before {gap:7.1f}"
    if prev_end:
        inside = [(x[2], (max(x[0], prev_end) - t0) / 1e3, (min(x[1], s) - t0) / 1e3) for x in tr if x[2] != nm and x[1] > prev_end and x[0] < s]
        line += "   in gap: " + ", ".join(f"{n}[{a:.0f}-{b:.0f}]" for n, a, b in inside[:8])
    print(line)
    prev_end = e
# all kernels in a window of 3 batches
w0, w1 = sel[4][0], sel[7][1]
print("\nwindow of three update launches (us from the first):")
for s, e, nm, q in tr:
    if e > w0 and s < w1:
        print(f"  {nm:28s} q={q:>3s} start {(s - w0) / 1e3:8.1f} end {(e - w0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f}")
durs = [(e - s) / 1e3 for s, e, _, _ in upd[8:]]
gaps = [(upd[i][0] - upd[i - 1][1]) / 1e3 for i in range(9, len(upd))]
gaps = [g for g in gaps if g < 2000]
print(f"\nupdate duration mean {sum(durs) / len(durs):.1f} us; gap between updates mean {sum(gaps) / len(gaps):.1f} us (n={len(gaps)})")
