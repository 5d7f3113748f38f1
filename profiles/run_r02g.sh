cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "specialised" 2>&1 | tail -3
{
for ab in 0 1 16 17; do echo "pc 12+4 ablate=$ab"; NGF_KERNEL=1 NGF_WAVES=124 NGF_ABLATE=$ab timeout 120 python profiles/workload.py triplane_R2 3; done
NGF_WAVES=124 timeout 200 python profiles/exp_sections_pc.py R2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02g_pc.txt
