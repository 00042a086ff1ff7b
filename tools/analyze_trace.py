"""Summarise one LML evaluation from a rocprofv3 kernel trace: per-stream busy time, gaps, overlap."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'migp' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# split into evaluations at assemble_kernel
starts = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']]
ev = rows[starts[-2]:starts[-1]] if len(starts) >= 2 else rows[starts[-1]:]
t0 = ev[0]['s']
print(f"eval span {(ev[-1]['e']-t0)/1e6:.2f} ms, {len(ev)} kernels")
byq = collections.defaultdict(list)
for r in ev: byq[r['Queue_Id']].append(r)
for q, lst in byq.items():
    busy = sum(r['e'] - r['s'] for r in lst)
    names = collections.Counter(r['Kernel_Name'].split('(')[0][-30:] for r in lst)
    print(f"queue {q}: {len(lst)} kernels busy {busy/1e6:.2f} ms; first {(lst[0]['s']-t0)/1e6:.2f} last end {(lst[-1]['e']-t0)/1e6:.2f}", dict(names))
# union busy of gemm vs non-gemm
def union(lst):
    tot = 0; cur_s = cur_e = None
    for r in sorted(lst, key=lambda r: r['s']):
        if cur_e is None or r['s'] > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = r['s'], r['e']
        else: cur_e = max(cur_e, r['e'])
    if cur_e is not None: tot += cur_e - cur_s
    return tot
g = [r for r in ev if 'gemm' in r['Kernel_Name']]
print(f"gemm union {union(g)/1e6:.2f} ms, all union {union(ev)/1e6:.2f} ms")
if len(sys.argv) > 2:
    for r in ev[:int(sys.argv[2])]:
        print(f"{(r['s']-t0)/1e3:9.1f} {(r['e']-r['s'])/1e3:8.1f} q{r['Queue_Id']} {r['Kernel_Name'][:50]} grid={r.get('Grid_Size','')}")
# timeline of main-queue (trailing) GEMMs vs panel chain per super-panel
big = [r for r in ev if 'gemm' in r['Kernel_Name'] and r['Queue_Id'] == ev[0]['Queue_Id']]
print("trailing GEMMs (start ms, dur us):")
for r in big[::2][:40]:
    print(f"  {(r['s']-t0)/1e6:7.2f} {(r['e']-r['s'])/1e3:8.1f}")
