"""Per-launch durations of the GEMM kernels of the LAST evaluation in a rocprofv3 kernel trace, in launch order:
python tools/trace_gemm_seq.py <kernel_trace.csv> [min_grid]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'migp' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']]
ev = rows[starts[-2]:starts[-1]]
ming = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tot = {}
for r in ev:
    nm = 'b' if 'kernel_b' in r['Kernel_Name'] else 's' if 'kernel_s' in r['Kernel_Name'] else None
    if nm is None: continue
    g = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0) // 256
    tot[nm] = tot.get(nm, 0) + (r['e'] - r['s'])
    if g >= ming:
        print(f"{(r['s']-ev[0]['s'])/1e6:8.3f} ms  {nm} wgs={g:6d} dur={(r['e']-r['s'])/1e3:8.1f} us")
print("eval span ms", (ev[-1]['e'] - ev[0]['s']) / 1e6, {k: v / 1e6 for k, v in tot.items()})
