cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_bf16" 2>&1 | tail -3
for w in 12 8; do for wl in triplane_R1_split triplane_R2_split triplane_R1_splitd; do echo "waves=$w"; NGF_WAVES=$w timeout 120 python profiles/workload.py $wl 8 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r02n_split_bf16.txt
for wl in triplane_R1 triplane_R1_bd; do NGF_WAVES=8 timeout 120 python profiles/workload.py $wl 8 2>&1 | grep -v amdgpu.ids; done
