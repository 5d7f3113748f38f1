cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -iE "^\s*(Name|Counter)?.*\b(TA_|TCP_|TD_|TCC_)" | head -400 > gpurun_out/r02c_counters.txt
wc -l gpurun_out/r02c_counters.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "specialised" 2>&1 | tail -5
{
for w in 8 12 16; do echo "fused waves=$w"; NGF_KERNEL=0 NGF_WAVES=$w timeout 120 python profiles/workload.py triplane_R0 5; NGF_KERNEL=0 NGF_WAVES=$w timeout 120 python profiles/workload.py triplane_R1 5; done
for w in 124 88 84; do echo "pc waves=$w"; for p in R0 R1 R2; do NGF_KERNEL=1 NGF_WAVES=$w timeout 120 python profiles/workload.py triplane_$p 5; done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02c_sweep.txt
OUT=gpurun_out/r02c_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  NGF_KERNEL=0 timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python profiles/workload.py triplane_R0 3 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python profiles/summarize_pmc.py $OUT gpurun_out/r02c_R0_fused_ta.json "ngf::render_kernel" > gpurun_out/r02c_R0_fused_ta.txt 2>&1
rm -rf $OUT
cat gpurun_out/r02c_R0_fused_ta.txt | head -40
