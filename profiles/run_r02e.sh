cd "$GRAFT_REPO_ROOT"
{ NGF_WAVES=124 timeout 200 python profiles/exp_sections_pc.py R1 R2; NGF_WAVES=88 timeout 200 python profiles/exp_sections_pc.py R1 R2; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02e_sections_pc.txt
