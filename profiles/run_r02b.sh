cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "specialised" 2>&1 | tail -15 > gpurun_out/r02b_pytest_pc.txt
cat gpurun_out/r02b_pytest_pc.txt
for k in 0 1; do NGF_KERNEL=$k timeout 120 python profiles/workload.py triplane_R1 10; NGF_KERNEL=$k timeout 120 python profiles/workload.py triplane_R2 5; NGF_KERNEL=$k timeout 120 python profiles/workload.py triplane_R0 5; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02b_pc_vs_fused.txt
