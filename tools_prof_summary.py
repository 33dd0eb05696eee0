"""Summarise rocprofv3 csv output (kernel stats + PMC per kernel) into a few lines."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


print("== kernel stats (tl3d + top 5) ==")
n = 0
for r in rows("trace/**/*kernel_stats.csv"):
    if "tl3d" in r.get("Name", "") or "rocclr" in r.get("Name", "") or n < 5:
        print(f"{r.get('Name','')[:64]:64s} calls={r.get('Calls'):>6s} avg_ns={float(r.get('AverageNs',0)):12.1f} min={r.get('MinNs')} max={r.get('MaxNs')} pct={r.get('Percentage')}")
    n += 1

agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in rows("pmc_*/**/*counter_collection.csv"):
    k = r.get("Kernel_Name", "")
    if "tl3d" not in k:
        continue
    k = k[:60]
    c = r.get("Counter_Name")
    agg[k][c] += float(r.get("Counter_Value", 0))
    cnt[k][c] += 1
print("== PMC (mean per dispatch, tl3d kernels) ==")
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print(f"    {c:32s} {agg[k][c] / max(1, cnt[k][c]):18.1f}  (n={cnt[k][c]})")
