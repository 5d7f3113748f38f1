mkdir -p gpurun_out/r04d
for tail in 0 8 16 32; do
  MODEL=infoinv TAIL=$tail SIZES="4096 80000 640000" python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu >> gpurun_out/r04d/infoinv_tail.txt
done
cat gpurun_out/r04d/infoinv_tail.txt
