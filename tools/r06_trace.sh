#!/bin/bash
# kernel timeline of the last evaluation at one size / option set: tools/r06_trace.sh N TAG [MIGP_OPTS]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; TAG=$2; export MIGP_OPTS=$3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tr_$TAG -- python3 $ROOT/tools/trace_n.py $N 8 lml > $ROOT/gpurun_out/tr_$TAG.log 2>&1
python3 $ROOT/tools/timeline.py $ROOT/gpurun_out/tr_$TAG > $ROOT/gpurun_out/r06_timeline_$TAG.txt 2>&1
rm -rf $ROOT/gpurun_out/tr_$TAG
tail -1 $ROOT/gpurun_out/tr_$TAG.log
