#!/bin/bash
# A/B of handle options on whole evaluations (interleaved in one process per size): tools/ab_opts.sh "<N d kernel>" set1 set2 ...
spec=$1; shift
python tools/dev_ab_opts.py $spec "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_ab.txt
