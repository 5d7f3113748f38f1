#!/bin/bash
# Round-end evidence run (on the GPU box, through gpurun):  bash profiles/collect_all.sh
#   1. default bench.py line                        -> gpurun_out/r01_bench.json
#   2. the SAME command under rocprofv3 --kernel-trace --stats -> gpurun_out/r01_kernel_stats_headline.txt (+ the bench line it printed)
#   3. PMC passes (own runs, kernel-trace only alongside) for the faithful and the fully baked variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python bench.py 2>gpurun_out/bench_stderr.log | grep '^{' > gpurun_out/r01_bench.json
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt -- python bench.py --extras 0 --cpu-seconds 0 2>/dev/null | grep '^{' > gpurun_out/r01_bench_headline_under_rocprof.json
python profiles/summarize_rocpd.py $(find gpurun_out/kt -name "*.db" | head -1) > gpurun_out/r01_kernel_stats_headline.txt
bash profiles/collect_pmc.sh faithful > /dev/null
bash profiles/collect_pmc.sh baked --bake-density 1 --bake-color 1 > /dev/null
python profiles/summarize_pmc.py gpurun_out/pmc_faithful gpurun_out/r01_pmc_faithful.json > gpurun_out/r01_pmc_faithful.txt
python profiles/summarize_pmc.py gpurun_out/pmc_baked gpurun_out/r01_pmc_baked.json > gpurun_out/r01_pmc_baked.txt
rm -rf gpurun_out/pmc_faithful/p*/ gpurun_out/pmc_baked/p*/ gpurun_out/kt
ls -la gpurun_out | head -30
