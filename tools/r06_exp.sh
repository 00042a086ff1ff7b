#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python tools/ab_lib.py "2048 8 RBF" "4096 8 RBF" "6144 8 RBF" "8192 8 RBF" "16384 16 Matern52" "8192 8 RBF grad" -- andvaranaut_amd/libmi_gp.so tools/ab/lib_occ3_68.so tools/ab/lib_occ3_70.so > gpurun_out/r06_h_occ3.txt 2>&1
grep median gpurun_out/r06_h_occ3.txt
