"""Per-stream accounting of ONE evaluation in a rocprofv3 kernel trace (round 4).

usage: python tools/trace_steps.py <rocprof output dir> [--leaves] [--gaps US] [--eval K]
Prints, for the K-th last evaluation (default: the last): span, busy / idle time of each HIP queue, the sum per kernel
class and queue, the idle gaps longer than --gaps microseconds with the kernels around them, and (--leaves) every
potrf_leaf128 launch with its duration and what ran on the OTHER queue while it ran (the leaf's
in-situ outliers)."""
import collections, csv, glob, sys

d = sys.argv[1]
show_leaves = "--leaves" in sys.argv
gap_us = float(sys.argv[sys.argv.index("--gaps") + 1]) if "--gaps" in sys.argv else 20.0
kth = int(sys.argv[sys.argv.index("--eval") + 1]) if "--eval" in sys.argv else 1
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "migp" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0].split("::")[-1]
    r["cls"] = ("gemm_b" if "kernel_b" in nm else "gemm_s" if "kernel_s" in nm else "leaf" if "potrf_leaf" in nm else
                "strip" if "trsm_strip" in nm else "assemble" if "assemble" in nm else nm[:20])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "set_yrows" in r["Kernel_Name"]]
i0 = starts[-kth]
i1 = starts[-kth + 1] if kth > 1 else len(rows)
ev = rows[i0:i1]
t0, t1 = ev[0]["s"], max(r["e"] for r in ev)
print(f"evaluation span {(t1 - t0) / 1e3:.1f} us, {len(ev)} kernels")
queues = collections.defaultdict(list)
for r in ev:
    queues[r["Queue_Id"]].append(r)
for q, rs in queues.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    first, last = rs[0]["s"], max(r["e"] for r in rs)
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e3:.1f} us, active window {(first - t0) / 1e3:.1f} .. {(last - t0) / 1e3:.1f} us, "
          f"idle inside the window {(last - first - busy) / 1e3:.1f} us")
    per = collections.Counter(); cnt = collections.Counter()
    for r in rs:
        per[r["cls"]] += r["e"] - r["s"]; cnt[r["cls"]] += 1
    for k, v in per.most_common():
        print(f"      {k:12s} n={cnt[k]:4d} sum={v / 1e3:9.1f} us avg={v / cnt[k] / 1e3:7.1f}")
    prev = None
    for r in rs:
        if prev is not None and (r["s"] - prev["e"]) / 1e3 >= gap_us:
            print(f"      gap {(r['s'] - prev['e']) / 1e3:7.1f} us at t={(prev['e'] - t0) / 1e3:8.1f}: after {prev['cls']} "
                  f"(dur {(prev['e'] - prev['s']) / 1e3:.1f}, grid {prev.get('Grid_Size_X', '')}) before {r['cls']} (grid {r.get('Grid_Size_X', '')})")
        prev = r
if show_leaves:
    print("leaves: t_start, duration, kernels of the other queue overlapping it")
    durs = []
    for r in ev:
        if r["cls"] != "leaf":
            continue
        ov = [o for o in ev if o["Queue_Id"] != r["Queue_Id"] and o["s"] < r["e"] and o["e"] > r["s"]]
        desc = ", ".join(f"{o['cls']}(grid {o.get('Grid_Size_X', '')}, {(min(o['e'], r['e']) - max(o['s'], r['s'])) / 1e3:.0f} us of {(o['e'] - o['s']) / 1e3:.0f})" for o in ov)
        durs.append((r["e"] - r["s"]) / 1e3)
        print(f"   t={(r['s'] - t0) / 1e3:8.1f} dur={(r['e'] - r['s']) / 1e3:6.1f} us | {desc or 'alone'}")
    import statistics
    print(f"leaf durations: n={len(durs)} mean {statistics.mean(durs):.1f} median {statistics.median(durs):.1f} max {max(durs):.1f} us; "
          f"> 100 us: {sum(1 for x in durs if x > 100)}")
