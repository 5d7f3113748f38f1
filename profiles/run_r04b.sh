set -x
mkdir -p gpurun_out/r04b
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04b/pytest_parity.txt 2>&1
tail -5 gpurun_out/r04b/pytest_parity.txt
python profiles/exp_launch_size.py > gpurun_out/r04b/launch_size.txt 2>&1
TILES="4 8" SIZES="4096 80000 160000" python profiles/exp_launch_size.py > gpurun_out/r04b/launch_size_tiles.txt 2>&1
NGF_LIB=neural-gauge-fields_amd/csrc/build/exp/timeline/libngf_hip.so python profiles/exp_timeline.py > gpurun_out/r04b/timeline.txt 2>&1
cat gpurun_out/r04b/launch_size.txt
