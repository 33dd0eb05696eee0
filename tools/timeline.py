"""Print the tl3d kernel timeline of a rocprofv3 kernel trace (start/end in us, queue) for a window of the run."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
tr = [r for r in csv.DictReader(open(f)) if "tl3d" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if "integrate_kernel<false" in r["Kernel_Name"]]
mid = idx[len(idx) // 3]
t0 = int(tr[mid - 8]["Start_Timestamp"])
for r in tr[mid - 8: mid + 20]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void tl3d::", "").replace("tl3d::", "")[:30]
    print(f"{name:30s} q={r['Queue_Id']} start={s/1e3:8.1f} end={e/1e3:8.1f} dur={(e-s)/1e3:6.1f}")
