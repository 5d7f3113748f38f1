#!/bin/bash
# Round-2 evidence run (on the GPU box, through gpurun):  bash profiles/collect_all_r02.sh
#   1. kernel-trace stats + PMC passes per workload (profiles/collect.sh) -> gpurun_out/r02_<workload>_{kernel_stats.txt,pmc.txt,pmc.json}
#   2. the default bench line                                         -> gpurun_out/r02_bench.json
#   3. the SAME command under rocprofv3 --kernel-trace --stats          -> gpurun_out/r02_kernel_stats_headline.txt
# Copy gpurun_out/r02_* into profiles/ afterwards (tracked).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for wl in triplane_R1 triplane_R0 triplane_R2 triplane_R1_bd triplane_R1_bdc triplane_R0_bd triplane_R2_bdc triplane_R1_nofold triplane_R1_split triplane_R2_split triplane_R1_splitd infoinv_R1 infoinv_R1_split; do
  bash profiles/collect.sh r02_$wl $wl "ngf::render_kernel" > /dev/null 2>&1
done
bash profiles/collect.sh r02_uv_sphere uv_sphere "uv_render_kernel" > /dev/null 2>&1
bash profiles/collect.sh r02_uv_sphere_split uv_sphere_split "uv_render_kernel" > /dev/null 2>&1
NGF_STAGE=1 bash profiles/collect.sh r02_triplane_R0_staged triplane_R0 "ngf::render_kernel" > /dev/null 2>&1
NGF_KERNEL=1 NGF_WAVES=88 bash profiles/collect.sh r02_triplane_R1_pc88 triplane_R1 "render_pc_kernel" > /dev/null 2>&1
# the bench line embeds the PMC summaries of THIS build (bench.py reads profiles/r02_<workload>_pmc.json and checks the .so hash)
cp gpurun_out/r02_*_pmc.json profiles/
timeout 900 python bench.py 2>gpurun_out/r02_bench.err | grep '^{' > gpurun_out/r02_bench.json
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt -- python bench.py --extras 0 --cpu-seconds 0 2>/dev/null | grep '^{' > gpurun_out/r02_bench_headline_under_rocprof.json
python profiles/summarize_rocpd.py $(find gpurun_out/kt -name "*.db" | head -1) > gpurun_out/r02_kernel_stats_headline.txt
rm -rf gpurun_out/kt
# training step: per-kernel times
rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > gpurun_out/r02_train_R1_kernel_stats.txt 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) >> gpurun_out/r02_train_R1_kernel_stats.txt
rm -rf gpurun_out/ktt
ls gpurun_out | grep r02_ | head -60
