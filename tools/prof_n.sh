#!/bin/bash
# usage: tools/prof_n.sh <tag> N d what   -> per-kernel stats of the last evaluations
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -- python3 $ROOT/tools/trace_n.py "$@" > $ROOT/gpurun_out/prof_${tag}.log 2>&1
tail -1 $ROOT/gpurun_out/prof_${tag}.log
f=$(find $ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
PY
