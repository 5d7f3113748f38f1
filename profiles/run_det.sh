mkdir -p gpurun_out/det
timeout 3000 python -m pytest tests/test_gpu_determinism.py -x -q -m gpu --durations=12 > gpurun_out/det/pytest_det.txt 2>&1
tail -25 gpurun_out/det/pytest_det.txt
