#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
{
timeout -k 10 200 python tools/dev_ab_opts.py 16384 16 Matern52 "47=0" "47=4" "47=2" || exit 1
timeout -k 10 200 python tools/dev_ab_opts.py 8192 8 RBF "47=0" "47=4" "47=2" || exit 1
timeout -k 10 200 python tools/dev_ab_opts.py 12288 8 RBF "47=0" "47=4" "47=2" || exit 1
timeout -k 10 200 python tools/dev_ab_opts.py 20480 8 RBF "47=0" "47=4" || exit 1
timeout -k 10 200 python tools/dev_ab_opts.py 6144 8 RBF "47=0" "47=2" || exit 1
} > gpurun_out/r06_first_w.txt 2>&1
grep -E "median|Error|error" gpurun_out/r06_first_w.txt
