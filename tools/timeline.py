"""Timeline of the last evaluation in a rocprofv3 kernel trace: python tools/timeline.py <dir> [max rows]
start (us since the evaluation's first kernel), duration, gap to the previous kernel's end on the same queue, queue, grid, kernel"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "migp" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "set_yrows" in r["Kernel_Name"]]
ev = rows[starts[-1]:]
t0 = ev[0]["s"]
last = {}
qs = {}
for r in ev[: int(sys.argv[2]) if len(sys.argv) > 2 else None]:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    gap = (r["s"] - last[q]) / 1e3 if q in last else 0.0
    last[q] = r["e"]
    name = r["Kernel_Name"].split("(")[0].replace("migp::", "").replace("void ", "")[:40]
    grid = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * (int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:7.1f} gap {gap:6.1f} q{q} wg {grid:5d}  {'    ' * q}{name}")
