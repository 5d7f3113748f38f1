# the whole GPU suite + the default bench line (what the driver runs at round end); outputs under gpurun_out/suite/
mkdir -p gpurun_out/suite
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/suite/pytest_gpu.txt 2>&1
tail -4 gpurun_out/suite/pytest_gpu.txt
python bench.py > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err
tail -c 3000 gpurun_out/suite/bench.json
