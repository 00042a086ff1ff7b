mkdir -p gpurun_out
O=gpurun_out/r05_sweep5.txt; rm -f $O
for N in 8192 12288; do
timeout -k 10 400 python tools/dev_ab_opts.py $N 8 RBF default "38=4" "38=2" "38=1" >> $O 2>&1
done
timeout -k 10 400 python tools/dev_ab_opts.py 16384 16 Matern52 default "38=4" "38=2" >> $O 2>&1
grep median $O
