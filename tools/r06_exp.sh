#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
{
timeout -k 10 200 python tools/dev_ab_opts.py 16384 16 Matern52 "24=1" "24=0" "24=1" "24=0 " || exit 1
timeout -k 10 200 python tools/dev_ab_opts.py 16384 8 RBF "24=1" "24=0" || exit 1
timeout -k 10 200 python tools/dev_ab_opts.py 14336 8 RBF "24=1" "24=0" || exit 1
timeout -k 10 200 python tools/dev_ab_opts.py 16384 16 Matern52 --grad "24=1" "24=0" || exit 1
timeout -k 10 200 python tools/dev_ab_opts.py 20480 8 RBF "24=1" "24=0" || exit 1
} > gpurun_out/r06_asm_split2.txt 2>&1
grep median gpurun_out/r06_asm_split2.txt
