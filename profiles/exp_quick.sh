#!/bin/bash
# quick A/B: the parity tests + the timed workloads (median of 4) of the main variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
for wl in ${WLS:-triplane_R1 triplane_R0 triplane_R2 triplane_R1_bd triplane_R1_split triplane_R1_splitd infoinv_R1}; do
  timeout 200 python profiles/workload.py $wl 4 2>/dev/null | grep -v amdgpu.ids
done
