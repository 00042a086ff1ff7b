#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
{
python tools/dev_ab_opts.py 16384 16 Matern52 "default" "4=64" "4=80" "4=96" "4=112"
python tools/dev_ab_opts.py 12288 8 RBF "default" "4=48" "4=64" "4=80"
} > gpurun_out/r06_g_opt4.txt 2>&1
cat gpurun_out/r06_g_opt4.txt
