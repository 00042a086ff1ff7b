#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_round.sh <tag>
# One call produces every rocprofv3 artefact a round commits under profiles/ (copy from gpurun_out/ afterwards):
#   <tag>_bench_line.json / <tag>_kernel_stats_bench.csv            bench.py default (look-ahead on)
#   <tag>_bench_line_nolookahead.json / ..._bench_nolookahead.csv   the dominant kernel alone on the chip
#   <tag>_kernel_stats_lml_grad_n16384.csv                          K7 kernels (U = L^-T levels, K^-1, contraction)
#   <tag>_kernel_stats_predict_10k_n16384.csv                       K8 kernels (cross-covariance, strip solves, reductions)
#   <tag>_kernel_stats_lml_n4096.csv / _n8192.csv                   the chain-bound sizes
#   gemm_traffic.json                                               HBM traffic of gemm_f64_kernel_b (two --pmc passes)
tag=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
stats() {  # dir -> csv copy
  f=$(find $OUT/$1 -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/$2
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_${tag}_bench -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded --chains-per-gpu 0 > $OUT/p_${tag}_bench.log 2>&1
grep '"metric"' $OUT/p_${tag}_bench.log > $OUT/${tag}_bench_line_profiled.json; stats p_${tag}_bench ${tag}_kernel_stats_bench.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_${tag}_nola -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded --chains-per-gpu 0 --no-lookahead --grad-steps 0 > $OUT/p_${tag}_nola.log 2>&1
grep '"metric"' $OUT/p_${tag}_nola.log > $OUT/${tag}_bench_line_nolookahead.json; stats p_${tag}_nola ${tag}_kernel_stats_bench_nolookahead.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_${tag}_grad -- python3 $ROOT/tools/trace_n.py 16384 16 grad > $OUT/p_${tag}_grad.log 2>&1
stats p_${tag}_grad ${tag}_kernel_stats_lml_grad_n16384.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_${tag}_pred -- python3 $ROOT/tools/trace_predict.py 16384 16 10000 > $OUT/p_${tag}_pred.log 2>&1
stats p_${tag}_pred ${tag}_kernel_stats_predict_10k_n16384.csv
for n in 4096 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_${tag}_n$n -- python3 $ROOT/tools/trace_n.py $n 8 lml > $OUT/p_${tag}_n$n.log 2>&1
  stats p_${tag}_n$n ${tag}_kernel_stats_lml_n$n.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --roofline-steps 1 --no-lookahead --no-cpu-baseline --no-sharded --chains-per-gpu 0 --grad-steps 0 > $OUT/pmc_$c.log 2>&1
done
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/gemm_traffic.json
for f in $OUT/p_${tag}_grad.log $OUT/p_${tag}_pred.log; do tail -n 2 $f | cut -c1-200; done
