cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python profiles/workload.py train_R1 20 2>&1 | grep -v amdgpu.ids
timeout 300 python profiles/workload.py train_R2 20 2>&1 | grep -v amdgpu.ids
