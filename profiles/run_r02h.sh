cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "specialised" 2>&1 | tail -3
{
for w in 124 88 84; do for ab in 0 16; do echo "pc waves=$w ablate=$ab"; for p in R0 R1 R2; do NGF_KERNEL=1 NGF_WAVES=$w NGF_ABLATE=$ab timeout 120 python profiles/workload.py triplane_$p 5; done; done; done
NGF_WAVES=124 timeout 200 python profiles/exp_sections_pc.py R1 R2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02h_pc.txt
