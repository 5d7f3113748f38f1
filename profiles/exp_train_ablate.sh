#!/bin/bash
# Where does train_density_bwd_kernel's time go?  NGF_ABLATE bits (timing only, gradients are wrong): 256 = no density-image atomics,
# 512 = no gauge-plane atomics; train_color_bwd_kernel: 65536 = no colour-plane atomics, 131072 = no colour-plane scatter at all
# (ABS="0 65536 131072"; wrong gradients change the next iterations' workload -- compare the MIN column).  Per-kernel figures from
# rocprofv3 --kernel-trace --stats.  In-kernel section clocks and transaction counts: profiles/exp_train_sections.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ab in ${ABS:-0 256 512 768}; do
  rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
  NGF_ABLATE=$ab timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > /dev/null 2>&1
  echo "NGF_ABLATE=$ab: $(python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name '*.db' | head -1) | grep -E "train_density_bwd|train_color_bwd")"
done
rm -rf gpurun_out/ktt
