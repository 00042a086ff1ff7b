#!/bin/bash
# round 6: the suite, fused strip + update on / off in one process, same-box A/B against round 5's library
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06_b_tests.txt 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r06_b_tests.txt
for N in 1024 2048 4096 8192; do python tools/dev_ab_opts.py $N 8 RBF "41=1" "41=0"; done > gpurun_out/r06_b_fuse.txt 2>&1
cat gpurun_out/r06_b_fuse.txt
python tools/ab_lib.py "2048 8 RBF" "4096 8 RBF" "8192 8 RBF" "16384 16 Matern52" -- tools/ab/r05/andvaranaut_amd/libmi_gp.so andvaranaut_amd/libmi_gp.so > gpurun_out/r06_b_ab.txt 2>&1
grep median gpurun_out/r06_b_ab.txt
echo done
