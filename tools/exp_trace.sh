ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT && bash tools/prof_round.sh r05 > $OUT/prof_round_r05.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p_r05_cols1 $OUT/p_r05_cols2
MIGP_OPTS=0=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/p_r05_cols1 -- python3 $ROOT/tools/trace_n.py 4096 8 lml > $OUT/p_r05_cols1.log 2>&1
python3 $ROOT/tools/column_times.py $OUT/p_r05_cols1 > $OUT/r05_column_times_single.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/p_r05_cols2 -- python3 $ROOT/tools/trace_n.py 4096 8 lml > $OUT/p_r05_cols2.log 2>&1
python3 $ROOT/tools/column_times.py $OUT/p_r05_cols2 > $OUT/r05_column_times_two.txt
python3 $ROOT/tools/timeline.py $OUT/p_r05_cols2 > $OUT/r05_timeline_n4096.txt
cd $ROOT && python bench.py > $OUT/r05_bench_final.log 2>&1
tail -2 $OUT/prof_round_r05.log | cut -c1-150; cat $OUT/r05_column_times_single.txt $OUT/r05_column_times_two.txt; grep '"metric"' $OUT/r05_bench_final.log | cut -c1-600
