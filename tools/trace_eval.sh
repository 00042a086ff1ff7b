#!/bin/bash
# kernel trace of a few LML evaluations at one size + per-stream accounting (tools/trace_steps.py)
N=${1:-16384}; D=${2:-16}; TAG=${3:-r04a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_$TAG -- python3 $ROOT/tools/trace_n.py $N $D lml > $ROOT/gpurun_out/trace_$TAG.log 2>&1
tail -1 $ROOT/gpurun_out/trace_$TAG.log
python3 $ROOT/tools/trace_steps.py $ROOT/gpurun_out/trace_$TAG --leaves --gaps 15 > $ROOT/gpurun_out/trace_${TAG}_steps.txt 2>&1
head -40 $ROOT/gpurun_out/trace_${TAG}_steps.txt
tail -3 $ROOT/gpurun_out/trace_${TAG}_steps.txt
