mkdir -p gpurun_out/parity
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/parity/pytest_parity.txt 2>&1
tail -15 gpurun_out/parity/pytest_parity.txt
