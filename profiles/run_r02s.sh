cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "infoinv_split" 2>&1 | tail -12
for wl in infoinv_R1 infoinv_R1_split; do timeout 120 python profiles/workload.py $wl 8 2>&1 | grep -v amdgpu.ids; done
