cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_bf16" 2>&1 | tail -12
for wl in triplane_R1 triplane_R1_split triplane_R2 triplane_R2_split triplane_R1_bd triplane_R1_splitd; do timeout 120 python profiles/workload.py $wl 8 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02m_split_bf16.txt
