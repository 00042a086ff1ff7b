#!/bin/bash
# soak of the final library: repeated / concurrent evaluations must return identical bits; handle life cycle (create probes the dispatch)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout -k 10 400 python tools/stress_repeat.py 30 > gpurun_out/r06_soak_repeat.txt 2>&1; echo "repeat rc=$?"; tail -2 gpurun_out/r06_soak_repeat.txt
timeout -k 10 300 python tools/stress_repeat.py 12 3 > gpurun_out/r06_soak_repeat3.txt 2>&1; echo "repeat x3 threads rc=$?"; tail -2 gpurun_out/r06_soak_repeat3.txt
timeout -k 10 300 python tools/stress_handles.py 3 > gpurun_out/r06_soak_handles.txt 2>&1; echo "handles rc=$?"; tail -1 gpurun_out/r06_soak_handles.txt
