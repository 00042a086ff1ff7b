"""Main-queue launches of the last evaluation of a rocprofv3 kernel trace with their rates (k = 1024 assumed), plus the
panel queue's first and last kernel per step: python tools/trace_main.py <rocprof dir> [tmin_us tmax_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
tmin = float(sys.argv[2]) if len(sys.argv) > 2 else -1
tmax = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
rows = [r for r in csv.DictReader(open(f)) if "migp" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
st = [i for i, r in enumerate(rows) if "set_yrows" in r["Kernel_Name"]]
ev = rows[st[-1]:]
t0, q1 = ev[0]["s"], ev[0]["Queue_Id"]
both = len(sys.argv) > 4
for r in ev:
    t = (r["s"] - t0) / 1e3
    if t < tmin or t > tmax:
        continue
    nm = r["Kernel_Name"].split("(")[0].split("::")[-1]
    nm = "B" if "kernel_b" in nm else "S" if "kernel_s" in nm else "leaf" if "leaf" in nm else "strip" if "strip" in nm else nm[:10]
    q = "M" if r["Queue_Id"] == q1 else "P"
    if q == "P" and not both:
        continue
    wg = int(r["Grid_Size_X"]) // 256
    dur = (r["e"] - r["s"]) / 1e3
    fl = wg * (2 * 128 * 128 * 1024 if nm == "B" else 2 * 64 * 64 * 1024) if nm in "BS" else 0
    print(f"{q} t={t:8.1f} end={(r['e'] - t0) / 1e3:8.1f} dur={dur:7.1f} {nm:6s} wgs={wg:5d} lds={r.get('LDS_Block_Size', '')} {fl / dur / 1e6 if fl else 0:5.1f} TF/s")
print(f"span {(max(r['e'] for r in ev) - t0) / 1e3:.1f} us")
