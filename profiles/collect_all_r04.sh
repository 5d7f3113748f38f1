#!/bin/bash
# Round-4 evidence run (on the GPU box, through gpurun):  bash profiles/collect_all_r04.sh [quick]
#   1. kernel-trace stats + PMC passes per workload (profiles/collect.sh) -> gpurun_out/r04_<workload>_{kernel_stats.txt,pmc.txt,pmc.json}
#   2. the default bench line (compact) + its side file                 -> gpurun_out/r04_bench.json, r04_bench_extras.json
#   3. the SAME command under rocprofv3 --kernel-trace --stats          -> gpurun_out/r04_kernel_stats_headline.txt
#   4. the training iteration per kernel, its timeline, one-stream A/B   -> gpurun_out/r04_train_R1_{kernel_stats,timeline,one_stream_kernel_stats,streams}.txt, r04_train_sections.txt
# Copy gpurun_out/r04_* into profiles/ afterwards (tracked).  Workload names: profiles/workload.py (`_bd` = level 2, the module default).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
WLS="triplane_R1_bd triplane_R1 triplane_R0_bd triplane_R2_bd triplane_R1_bdc triplane_R1_splitd infoinv_R1 infoinv_R1_split"
[ "$1" = quick ] && WLS="triplane_R1_bd"
for wl in $WLS; do
  bash profiles/collect.sh r04_$wl $wl "ngf::render_kernel" > /dev/null 2>&1
done
if [ "$1" != quick ]; then
  bash profiles/collect.sh r04_uv_sphere uv_sphere "uv_render_kernel" > /dev/null 2>&1
  bash profiles/collect.sh r04_uv_sphere_split uv_sphere_split "uv_render_kernel" > /dev/null 2>&1
fi
# the bench line embeds the PMC summaries of THIS build (bench.py reads profiles/r04_<workload>_pmc.json and checks the .so hash)
cp gpurun_out/r04_*_pmc.json profiles/
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r04_bench.err | grep '^{' > gpurun_out/r04_bench.json
cp bench_extras.json gpurun_out/r04_bench_extras.json
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt -- python bench.py --steps 20 --warmup 5 --extras 0 --cpu-seconds 0 2>/dev/null | grep '^{' > gpurun_out/r04_bench_headline_under_rocprof.json
python profiles/summarize_rocpd.py $(find gpurun_out/kt -name "*.db" | head -1) > gpurun_out/r04_kernel_stats_headline.txt
rm -rf gpurun_out/kt
if [ "$1" != quick ]; then
  # the training iteration: per-kernel times and the timeline of one iteration (the step's three streams), then the same on ONE stream
  # (ngf_debug_set("ablate", 1 << 19): kernel times then add up to the iteration) and the colour backward's / scatter's section clocks
  rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > gpurun_out/r04_train_R1_kernel_stats.txt 2>/dev/null
  python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) >> gpurun_out/r04_train_R1_kernel_stats.txt
  python profiles/timeline_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) > gpurun_out/r04_train_R1_timeline.txt 2>&1
  rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
  NGF_ABLATE=524288 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > gpurun_out/r04_train_R1_one_stream_kernel_stats.txt 2>/dev/null
  python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) >> gpurun_out/r04_train_R1_one_stream_kernel_stats.txt
  rm -rf gpurun_out/ktt
  for rep in 1 2 3; do for a in 0 524288; do echo -n "ablate=$a ($([ $a = 0 ] && echo 'three streams' || echo 'one stream')): "; NGF_ABLATE=$a timeout 120 python profiles/workload.py train_R1 20 2>&1 | grep '^train_R1'; done; done > gpurun_out/r04_train_R1_streams.txt
  timeout 120 python profiles/exp_train_sections.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_train_sections.txt
fi
# launch size vs time of the product library (level 2): the table VERDICT r3 asked for (4 096 / 40 000 / 80 000 / 160 000 / 640 000 rays), InfoInv beside it
python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_shard_latency.txt
LEVEL=2 python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_shard_latency.txt
MODEL=infoinv python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_shard_latency.txt
ls gpurun_out | grep r04_ | head -80
