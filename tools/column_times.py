"""Column times of the panel chain from a rocprofv3 kernel trace of single-stream evaluations (MIGP_OPTS=0=0):
leaf start -> next leaf start, and the kernels in between, for the last evaluation.
    python tools/column_times.py <rocprof output dir>"""
import csv, glob, sys
import numpy as np

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "migp" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "set_yrows" in r["Kernel_Name"]]
ev = rows[starts[-1]:]
leaves = [i for i, r in enumerate(ev) if "potrf_leaf" in r["Kernel_Name"]]
cols, parts = [], {"leaf": [], "strip": [], "update": [], "gaps": []}
for a, b in zip(leaves[:-1], leaves[1:]):
    seg = ev[a:b]
    cols.append((ev[b]["s"] - ev[a]["s"]) / 1e3)
    busy = {"leaf": 0.0, "strip": 0.0, "update": 0.0}
    for r in seg:
        k = "leaf" if "potrf_leaf" in r["Kernel_Name"] else "strip" if "trsm_strip" in r["Kernel_Name"] else "update"
        busy[k] += (r["e"] - r["s"]) / 1e3
    for k in busy:
        parts[k].append(busy[k])
    parts["gaps"].append(cols[-1] - sum(busy.values()))
cols = np.array(cols)
print(f"evaluation: {(max(r['e'] for r in ev) - ev[0]['s']) / 1e3:.1f} us, {len(leaves)} tile columns")
print(f"column time (leaf start -> next leaf start): median {np.median(cols):.1f} us, mean {cols.mean():.1f}, min {cols.min():.1f}, max {cols.max():.1f}")
for k, v in parts.items():
    print(f"   {k:7s} per column: median {np.median(v):6.1f} us, mean {np.mean(v):6.1f}")
print("columns (us):", " ".join(f"{c:.0f}" for c in cols))
