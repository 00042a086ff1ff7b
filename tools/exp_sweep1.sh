set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lml.py tests/test_gpu_grad_predict.py tests/test_gpu_gemm.py -x -q -m gpu > gpurun_out/r05_swz_tests.log 2>&1 || { tail -30 gpurun_out/r05_swz_tests.log; exit 1; }
tail -2 gpurun_out/r05_swz_tests.log
timeout -k 10 600 python tools/ab_lib.py "2048 8 RBF" "4096 8 RBF" "8192 8 RBF" -- tools/ab/lib_a.so tools/ab/lib_b.so > gpurun_out/r05_swz_ab.txt 2>&1
cat gpurun_out/r05_swz_ab.txt
