"""Timeline of the last evaluation in a rocprofv3 kernel trace: start, duration, queue, gap to the previous kernel of the queue."""
import csv, glob, sys, collections
d = sys.argv[1]
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'migp' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']]
ev = rows[starts[-1]:]
t0 = ev[0]['s']
print(f"eval span {(ev[-1]['e']-t0)/1e3:.1f} us, {len(ev)} kernels")
tot = collections.Counter(); cnt = collections.Counter()
for r in ev:
    nm = r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
    tot[nm] += r['e'] - r['s']; cnt[nm] += 1
for k, v in tot.items():
    print(f"   {k:30s} n={cnt[k]:4d} sum={v/1e3:8.1f} us avg={v/cnt[k]/1e3:6.1f}")
prev = {}
for r in ev[:nmax]:
    nm = r['Kernel_Name'].split('(')[0].split('::')[-1][:24]
    q = r['Queue_Id']
    gap = (r['s'] - prev.get(q, r['s'])) / 1e3
    prev[q] = r['e']
    print(f"   t={(r['s']-t0)/1e3:8.1f} dur={(r['e']-r['s'])/1e3:7.1f} q{q} gap={gap:6.1f} {nm} grid={r.get('Grid_Size_X', r.get('Grid_Size',''))}")
