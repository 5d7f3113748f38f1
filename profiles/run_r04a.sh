set -x
mkdir -p gpurun_out/r04a
python profiles/exp_launch_size.py > gpurun_out/r04a/launch_size_base.txt 2>&1
TILES="4 8" SIZES="4096 80000 160000" python profiles/exp_launch_size.py > gpurun_out/r04a/launch_size_tiles.txt 2>&1
NGF_LIB=neural-gauge-fields_amd/csrc/build/exp/timeline/libngf_hip.so python profiles/exp_timeline.py > gpurun_out/r04a/timeline.txt 2>&1
tail -30 gpurun_out/r04a/*.txt
