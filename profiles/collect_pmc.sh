#!/bin/bash
# PMC passes for the render kernel (run on the GPU box through gpurun).  One counter group per pass
# (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only alongside.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1; shift
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --extras 0 --cpu-seconds 0 $@"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed"
done
find $OUT -name "*counter_collection.csv" | head -20
