#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests/test_gpu_stream_edges.py -m gpu -q -k "serialised or demotes or hook or batch_with" > gpurun_out/r06_c_tests.txt 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r06_c_tests.txt
{
python tools/dev_ab_opts.py 2048 8 RBF "42=1" "42=2" "42=3" "43=1"
python tools/dev_ab_opts.py 4096 8 RBF "42=1" "42=2" "42=3" "42=4" "43=1" "37=32" "37=32,42=2" "37=32,42=3" "37=32,42=4" "37=32,43=1"
python tools/dev_ab_opts.py 8192 8 RBF "42=1" "42=2" "42=4" "43=1" "37=32,42=2" "37=40,42=4" "37=32"
} > gpurun_out/r06_c_group.txt 2>&1
cat gpurun_out/r06_c_group.txt
