cd "$GRAFT_REPO_ROOT"
{
for ab in 0 4 2 6 16; do echo "pc 12+4 ablate=$ab"; NGF_KERNEL=1 NGF_WAVES=124 NGF_ABLATE=$ab timeout 120 python profiles/workload.py triplane_R2 3; done
for ab in 0 4 2 6; do echo "pc 8+8 ablate=$ab"; NGF_KERNEL=1 NGF_WAVES=88 NGF_ABLATE=$ab timeout 120 python profiles/workload.py triplane_R2 3; done
for ab in 0 4 2 6; do echo "fused ablate=$ab"; NGF_KERNEL=0 NGF_ABLATE=$ab timeout 120 python profiles/workload.py triplane_R2 3; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02d_ablate.txt
