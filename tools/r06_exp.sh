#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests/test_gpu_grad_predict.py tests/test_gpu_batch.py tests/test_gpu_register_poison.py -m gpu -q -x > gpurun_out/r06_k_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r06_k_tests.txt
python tools/ab_lib.py "4096 8 RBF grad" "8192 8 RBF grad" "16384 16 Matern52 grad" "8192 8 RBF+Matern32 grad" "8192 8 RatQuad grad" -- tools/ab/r05/andvaranaut_amd/libmi_gp.so andvaranaut_amd/libmi_gp.so tools/ab/lib_gc4b.so > gpurun_out/r06_k_gc.txt 2>&1
grep median gpurun_out/r06_k_gc.txt
