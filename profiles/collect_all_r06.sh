#!/bin/bash
# Round-6 evidence run (on the GPU box, through gpurun):  bash profiles/collect_all_r06.sh [quick]
#   1. kernel-trace stats + PMC passes per workload (profiles/collect.sh) -> gpurun_out/r06_<workload>_{kernel_stats.txt,pmc.txt,pmc.json}
#      EVERY preset at the module's default level (3: R0 / R1 / R2 _bdc), levels 2 / 1 / 0 of the headline frame, the opt-in bf16 forms (level 2 _splitd,
#      level 3 _bdcs), the reference's own evaluation shape (S = 884 + alpha mask: _S884mask, _S884ball), InfoInv, UV-Mapping
#   2. the default bench line (compact) + its side file                 -> gpurun_out/r06_bench.json, r06_bench_extras.json
#   3. the SAME command under rocprofv3 --kernel-trace --stats          -> gpurun_out/r06_kernel_stats_headline.txt
#   4. the training iteration per kernel (three streams / one stream)    -> gpurun_out/r06_train_R1_{kernel_stats,one_stream_kernel_stats,streams}.txt
#   5. launch size vs time (one rank's shard, the reference's chunk)     -> gpurun_out/r06_shard_latency.txt
# Copy gpurun_out/r06_* into profiles/ afterwards (tracked).  Workload names: profiles/workload.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
WLS="triplane_R1_bdc triplane_R0_bdc triplane_R2_bdc triplane_R1_bd triplane_R1 triplane_R1_nofold triplane_R1_splitd triplane_R1_bdcs triplane_R2_bd triplane_R1_bdc_S884mask triplane_R2_bdc_S884mask triplane_R1_bdc_S884ball infoinv_R1 infoinv_R1_split triplane_R1_split triplane_R0 triplane_R2_splitd triplane_R2_bdcs infoinv_R1__S884mask infoinv_R1__S884ball triplane_R1_bdc_S884lattice"
[ "$1" = quick ] && WLS="triplane_R1_bdc"
for wl in $WLS; do
  bash profiles/collect.sh r06_$wl $wl "ngf::render_kernel" > /dev/null 2>&1
done
if [ "$1" != quick ]; then
  bash profiles/collect.sh r06_uv_sphere uv_sphere "uv_render_kernel" > /dev/null 2>&1
  bash profiles/collect.sh r06_uv_sphere_split uv_sphere_split "uv_render_kernel" > /dev/null 2>&1
fi
# the bench line embeds the PMC summaries of THIS build (bench.py reads profiles/r06_<workload>_pmc.json and checks the .so hash)
cp gpurun_out/r06_*_pmc.json profiles/
timeout 1200 python bench.py --steps 20 --warmup 5 2>gpurun_out/r06_bench.err | grep '^{' > gpurun_out/r06_bench.json
cp bench_extras.json gpurun_out/r06_bench_extras.json
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt -- python bench.py --steps 20 --warmup 5 --extras 0 --cpu-seconds 0 2>/dev/null | grep '^{' > gpurun_out/r06_bench_headline_under_rocprof.json
python profiles/summarize_rocpd.py $(find gpurun_out/kt -name "*.db" | head -1) > gpurun_out/r06_kernel_stats_headline.txt
rm -rf gpurun_out/kt
if [ "$1" != quick ]; then
  rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > gpurun_out/r06_train_R1_kernel_stats.txt 2>/dev/null
  python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) >> gpurun_out/r06_train_R1_kernel_stats.txt
  rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
  NGF_ABLATE=524288 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > gpurun_out/r06_train_R1_one_stream_kernel_stats.txt 2>/dev/null
  python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name "*.db" | head -1) >> gpurun_out/r06_train_R1_one_stream_kernel_stats.txt
  rm -rf gpurun_out/ktt
  for rep in 1 2 3; do for a in 0 524288; do echo -n "ablate=$a ($([ $a = 0 ] && echo 'three streams' || echo 'one stream')): "; NGF_ABLATE=$a timeout 120 python profiles/workload.py train_R1 20 2>&1 | grep '^train_R1'; done; done > gpurun_out/r06_train_R1_streams.txt
fi
SIZES="2000 4096 8000 16000 40000 80000 160000 640000" python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_shard_latency.txt
LEVEL=2 python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_shard_latency.txt
MODEL=infoinv python profiles/exp_launch_size.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_shard_latency.txt
if [ "$1" != quick ]; then
  timeout 600 python profiles/exp_rank_shards.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_rank_shards.txt
  timeout 600 python profiles/exp_pipeline_gap.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_pipeline_gap.txt
  # one render stream against two alternating ones (what bench.py runs at N > 1), 80 000-ray shards and whole frames; tail knob under both
  (TAILS=8,12,16,24 timeout 600 python profiles/exp_two_streams.py; SHARDS=1 timeout 600 python profiles/exp_two_streams.py) 2>&1 | grep -v "amdgpu.ids\|socket\|version\|Hostname\|Librccl" > gpurun_out/r06_two_streams.txt
fi
if [ "$1" != quick ]; then
  # an exchange kernel that NEEDS CUs behind every frame (stand-in for RCCL's all-gather on a real node), four against eight hardware queues
  [ -f profiles/micro/libcu_holder.so ] || hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o profiles/micro/libcu_holder.so profiles/micro/cu_holder.hip
  for q in 4 8; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q STREAMS=1 timeout 300 python profiles/exp_exchange_contention.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_exchange_contention_streams.txt
fi
python profiles/exp_autograd_loop.py both 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_autograd_loop.txt
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r06_pytest_gpu.txt
ls gpurun_out | grep r06_ | head -100
