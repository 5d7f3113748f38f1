#!/bin/bash
# Evidence for one workload (on the GPU box, through gpurun):   bash profiles/collect.sh <tag> <workload> [kernel-substring]
#   1. rocprofv3 --kernel-trace --stats of `python profiles/workload.py <workload>`   -> gpurun_out/<tag>_kernel_stats.txt
#   2. PMC passes, one counter group per run, kernel-trace only alongside (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE
#      costs 3, WRITE_SIZE 2; SQ 8 slots)                                             -> gpurun_out/<tag>_pmc.txt / .json
# Copy the summaries you keep into profiles/ (tracked).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; WL=$2; KSUB=${3:-ngf::render_kernel}
OUT=gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
CMD="python profiles/workload.py $WL 4"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1 || echo "kernel-trace run failed"
grep "^$WL\|^uv_sphere\|^train" $OUT/kt.log > gpurun_out/${TAG}_kernel_stats.txt
python profiles/summarize_rocpd.py $(find $OUT/kt -name "*.db" | head -1) >> gpurun_out/${TAG}_kernel_stats.txt 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed"
done
python profiles/summarize_pmc.py $OUT gpurun_out/${TAG}_pmc.json "$KSUB" > gpurun_out/${TAG}_pmc.txt 2>&1
rm -rf $OUT
tail -30 gpurun_out/${TAG}_pmc.txt
