set -x
./tools/probe_math > gpurun_out/r03d_probe_math.txt 2>&1; cat gpurun_out/r03d_probe_math.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r03d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03d_pytest.log; tail -6 gpurun_out/r03d_pytest.log
python tools/emulate_rank.py --curve --out gpurun_out/r03d_sharded_model.json > gpurun_out/r03d_emul.log 2>&1; grep -v '^{"world".*steps' gpurun_out/r03d_emul.log | cut -c1-700 | tail -30
