#!/bin/bash
# round 6 final measurements, part 2: the suite, the bench line (with this round's PMC traffic record), chain stamps, full-size
# gradient parity, configs 3 and 5 through the host drivers, the sharded model
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06_final_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06_final_tests.txt
python bench.py > gpurun_out/r06_bench_stdout.txt 2> gpurun_out/r06_bench_stderr.txt; grep '"metric"' gpurun_out/r06_bench_stdout.txt > gpurun_out/r06_bench_line.json; cut -c1-200 gpurun_out/r06_bench_line.json
( cd /tmp && export TMPDIR=/tmp && MIGP_OPTS=0=0 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tr_cs -- python3 $ROOT/tools/trace_n.py 4096 8 lml > $ROOT/gpurun_out/tr_cs.log 2>&1 )
{ echo "N=4096 RBF d=8, ONE stream (MIGP_OPTS=0=0), rocprofv3 kernel trace, last evaluation (tools/column_times.py):"; python tools/column_times.py gpurun_out/tr_cs; } > gpurun_out/r06_chain_stamps.txt 2>&1
rm -rf gpurun_out/tr_cs; cat gpurun_out/r06_chain_stamps.txt
python tools/fullsize_parity.py c3grad --out gpurun_out/r06_fullsize_parity.json > gpurun_out/r06_fullsize.log 2>&1; tail -2 gpurun_out/r06_fullsize.log | cut -c1-300
python tools/run_configs.py > gpurun_out/r06_configs_3_5.txt 2>&1; tail -3 gpurun_out/r06_configs_3_5.txt | cut -c1-300
python tools/emulate_rank.py --curve --out gpurun_out/r06_sharded_model.json > gpurun_out/r06_emulate.log 2>&1; grep predicted_ms gpurun_out/r06_emulate.log | cut -c1-220 | tail -12
