set -x
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_configs.py tests/test_gpu_gemm.py -x -q > gpurun_out/r03c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03c_pytest.log; tail -15 gpurun_out/r03c_pytest.log
./tools/probe_rcp > gpurun_out/r03c_probe_rcp.txt 2>&1; cat gpurun_out/r03c_probe_rcp.txt
python tools/dev_ab_opts.py 16384 16 Matern52 default 7=768 7=512 7=448 7=384 0=0 0=0,7=512 > gpurun_out/r03c_ab7_n16384.txt 2>&1; cat gpurun_out/r03c_ab7_n16384.txt
python tools/dev_ab_opts.py 8192 8 RBF default 7=768 7=512 7=448 7=384 > gpurun_out/r03c_ab7_n8192.txt 2>&1; cat gpurun_out/r03c_ab7_n8192.txt
python tools/dev_ab_opts.py 4096 8 RBF default 7=512 7=448 7=384 7=256 > gpurun_out/r03c_ab7_n4096.txt 2>&1; cat gpurun_out/r03c_ab7_n4096.txt
python tools/emulate_rank.py --n 16384 --d 16 --kernel Matern52 --world 1 --ranks 0 --out gpurun_out/r03c_emul_n16384_w1.json > gpurun_out/r03c_emul.log 2>&1; tail -4 gpurun_out/r03c_emul.log | cut -c1-600
