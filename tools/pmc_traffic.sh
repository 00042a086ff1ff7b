#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters (run on the GPU box from the repo root).
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC slot budget, MI355X_MICROARCH.md rocprofv3 PMC slots).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --roofline-steps 1 --no-cpu-baseline > $ROOT/gpurun_out/pmc_$c.log 2>&1
done
python3 $ROOT/tools/pmc_traffic.py $ROOT/gpurun_out/pmc_FETCH_SIZE $ROOT/gpurun_out/pmc_WRITE_SIZE $ROOT/gpurun_out/gemm_traffic.json
