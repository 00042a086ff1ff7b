#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python tools/ab_lib.py "1024 8 RBF" "2048 8 RBF" "4096 8 RBF" "8192 8 RBF" -- andvaranaut_amd/libmi_gp.so tools/ab/lib_leafpub.so > gpurun_out/r06_o_leafpub.txt 2>&1
grep median gpurun_out/r06_o_leafpub.txt
