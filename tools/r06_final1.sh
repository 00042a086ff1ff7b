#!/bin/bash
# round 6 final measurements, part 1: rocprofv3 summaries (tools/prof_round.sh), the default bench line, same-box A/B against round 5
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/prof_round.sh r06 > gpurun_out/r06_prof_round.log 2>&1; tail -3 gpurun_out/r06_prof_round.log | cut -c1-200
python bench.py > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench_stderr.txt; grep '"metric"' gpurun_out/r06_bench_stdout.txt > gpurun_out/r06_bench_line.json; cut -c1-400 gpurun_out/r06_bench_line.json
python tools/ab_lib.py "1024 8 RBF" "2048 8 RBF" "3072 8 RBF" "4096 8 RBF" "6144 8 RBF" "8192 8 RBF" "12288 8 RBF" "16384 16 Matern52" -- tools/ab/r05/andvaranaut_amd/libmi_gp.so andvaranaut_amd/libmi_gp.so > gpurun_out/r06_vs_r05_ab.txt 2>&1
python tools/ab_lib.py "2048 8 RBF grad" "4096 8 RBF grad" "8192 8 RBF grad" "16384 16 Matern52 grad" -- tools/ab/r05/andvaranaut_amd/libmi_gp.so andvaranaut_amd/libmi_gp.so > gpurun_out/r06_vs_r05_ab_grad.txt 2>&1
grep median gpurun_out/r06_vs_r05_ab.txt gpurun_out/r06_vs_r05_ab_grad.txt
