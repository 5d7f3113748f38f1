#!/bin/bash
# PMC passes for the UV-Mapping kernel (76 800 rays x 64 samples, sphere gauge)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_uv; mkdir -p $OUT
cat > /tmp/uv_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import torch, ngf_amd
from ngf_amd import synth, uvmapping
net = uvmapping.NeuTex(primitive_type="sphere", sample_num=64, device="cuda"); net.load_params(synth.uvmapping_params(5, "sphere"))
cam, dirs = synth.dtu_rays(600, 800, rows=(252, 348))
cam_t, dirs_t = torch.from_numpy(cam)[None], torch.from_numpy(dirs)[None].cuda()
U = torch.rand((1, dirs.shape[0], 64), device="cuda")
for _ in range(2): net(cam_t, dirs_t, None, jitter_u=U)
torch.cuda.synchronize()
PY
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python /tmp/uv_run.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
