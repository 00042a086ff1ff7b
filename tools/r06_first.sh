#!/bin/bash
# round 6, first GPU call: the suite, same-box A/B against round 5's library, kernel timelines of N = 4096 / 8192
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r06_a_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r06_a_tests.txt
python tools/ab_lib.py "4096 8 RBF" "8192 8 RBF" "16384 16 Matern52" -- tools/ab/lib_r05.so andvaranaut_amd/libmi_gp.so > gpurun_out/r06_a_ab.txt 2>&1
cat gpurun_out/r06_a_ab.txt
cd /tmp && export TMPDIR=/tmp
for N in 4096 8192; do
  rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r06_a_trace_$N -- python3 $ROOT/tools/trace_n.py $N 8 lml > $ROOT/gpurun_out/r06_a_trace_$N.log 2>&1
  python3 $ROOT/tools/timeline.py $ROOT/gpurun_out/r06_a_trace_$N > $ROOT/gpurun_out/r06_a_timeline_$N.txt 2>&1
  rm -rf $ROOT/gpurun_out/r06_a_trace_$N
done
echo done
