cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_uv.py tests/test_gpu_uv_edit.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -12
for wl in uv_sphere uv_sphere_split; do timeout 200 python profiles/workload.py $wl 6 2>&1 | grep -v amdgpu.ids; done
