mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full_gpu4.log 2>&1; tail -5 gpurun_out/r05_full_gpu4.log
O=gpurun_out/r05_col3.txt; rm -f $O
timeout -k 10 200 python tools/dev_ab_opts.py 4096 8 RBF --grad "37=0" "37=24" >> $O 2>&1
timeout -k 10 200 python tools/dev_ab_opts.py 8192 8 RBF --grad "37=0" "37=24" >> $O 2>&1
timeout -k 10 300 python tools/dev_ab_opts.py 16384 16 Matern52 --grad "37=0" "37=24" >> $O 2>&1
grep median $O
