ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p_thin_n16384
rocprofv3 --kernel-trace --output-format csv -d $OUT/p_thin_n16384 -- python3 $ROOT/tools/trace_n.py 16384 16 lml > $OUT/p_thin_n16384.log 2>&1
python3 $ROOT/tools/timeline.py $OUT/p_thin_n16384 > $OUT/p_thin_n16384_timeline.txt
tail -1 $OUT/p_thin_n16384.log
