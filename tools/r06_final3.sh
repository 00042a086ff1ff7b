#!/bin/bash
# round 6, after the assembly split: the suite on the final library, the LML A/B against round 5 again, kernel statistics of N = 4096
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06_final_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06_final_tests.txt
python tools/ab_lib.py "512 4 RBF" "1024 8 RBF" "2048 8 RBF" "3072 8 RBF" "4096 8 RBF" "6144 8 RBF" "8192 8 RBF" "12288 8 RBF" "16384 16 Matern52" -- tools/ab/r05/andvaranaut_amd/libmi_gp.so andvaranaut_amd/libmi_gp.so > gpurun_out/r06_vs_r05_ab.txt 2>&1
grep median gpurun_out/r06_vs_r05_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/p_r06_n4096 -- python3 $ROOT/tools/trace_n.py 4096 8 lml > $ROOT/gpurun_out/p_r06_n4096.log 2>&1
f=$(find $ROOT/gpurun_out/p_r06_n4096 -name "*kernel_stats.csv" | head -1); cp "$f" $ROOT/gpurun_out/r06_kernel_stats_lml_n4096.csv; rm -rf $ROOT/gpurun_out/p_r06_n4096
grep -i "assemble" $ROOT/gpurun_out/r06_kernel_stats_lml_n4096.csv | cut -c1-60,200-300
