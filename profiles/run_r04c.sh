set -x
mkdir -p gpurun_out/r04c
rm -f gpurun_out/r04c/launch_size_tail.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04c/pytest_parity.txt 2>&1
tail -5 gpurun_out/r04c/pytest_parity.txt
for tail in 0 8 16 24 32 48; do
  TAIL=$tail python profiles/exp_launch_size.py >> gpurun_out/r04c/launch_size_tail.txt 2>&1
done
NGF_LIB=neural-gauge-fields_amd/csrc/build/exp/timeline/libngf_hip.so python profiles/exp_timeline.py > gpurun_out/r04c/timeline.txt 2>&1
grep -v "rows from" gpurun_out/r04c/launch_size_tail.txt | grep -v amdgpu
