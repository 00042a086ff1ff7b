#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [bench args...]
# writes gpurun_out/prof_<tag>/ and prints the per-kernel stats
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $ROOT/gpurun_out/prof_${tag}_bench.log 2>&1
grep '"metric"' $ROOT/gpurun_out/prof_${tag}_bench.log | cut -c1-200
f=$(find $ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
PY
