cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python bench.py --steps 20 --warmup 5 2>gpurun_out/r06_bench.err | grep '^{' > gpurun_out/r06_bench.json
cp bench_extras.json gpurun_out/r06_bench_extras.json
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt -- python bench.py --steps 20 --warmup 5 --extras 0 --cpu-seconds 0 2>/dev/null | grep '^{' > gpurun_out/r06_bench_headline_under_rocprof.json
python profiles/summarize_rocpd.py $(find gpurun_out/kt -name "*.db" | head -1) > gpurun_out/r06_kernel_stats_headline.txt
rm -rf gpurun_out/kt
head -4 gpurun_out/r06_kernel_stats_headline.txt | cut -c1-200
cut -c1-200 gpurun_out/r06_bench.json
