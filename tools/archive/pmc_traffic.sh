#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_traffic.sh  -> gpurun_out/gemm_traffic.json (copy to profiles/)
# The two PMC passes of tools/prof_round.sh alone: HBM bytes per launch of gemm_f64_kernel_b and assemble_kernel.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --roofline-steps 1 --no-cpu-baseline --no-sharded --chains-per-gpu 0 --grad-steps 0 > $OUT/pmc_$c.log 2>&1
done
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/gemm_traffic.json
