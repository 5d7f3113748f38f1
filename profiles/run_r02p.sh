cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for wl in triplane_R1 triplane_R2 triplane_R1_split triplane_R1_splitd infoinv_R1; do timeout 120 python profiles/workload.py $wl 8 2>&1 | grep -v amdgpu.ids; done
