# A/B of two builds of the library on ONE box, alternating: MODEL / SIZES as exp_launch_size.py; LIBS = "path path"
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
  for lib in $LIBS; do
    echo "== $lib" >> gpurun_out/ab/$TAG.txt
    NGF_LIB=$lib REP=20 python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu | grep -v "rows from" >> gpurun_out/ab/$TAG.txt
  done
done
cat gpurun_out/ab/$TAG.txt
