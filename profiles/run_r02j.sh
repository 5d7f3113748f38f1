cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "staged" 2>&1 | tail -5
timeout 600 python profiles/exp_lds_staging.py R0 R1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02j_lds_staging.txt
