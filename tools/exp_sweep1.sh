mkdir -p gpurun_out
timeout -k 10 1000 python tools/ab_lib.py "2048 8 RBF grad" "4096 8 RBF grad" "8192 8 RBF grad" "16384 16 Matern52 grad" -- tools/ab/lib_r4.so tools/ab/lib_r5.so > gpurun_out/r05_vs_r04_ab_grad.txt 2>&1
cat gpurun_out/r05_vs_r04_ab_grad.txt
