#!/bin/bash
# Where does train_density_bwd_kernel's time go?  NGF_ABLATE bits (timing only, gradients are wrong): 256 = no density-image scatter,
# 512 = no gauge-plane scatter (wrong gradients change the next iterations' workload -- compare the MIN column).  Bit 524288 (1 << 19) is
# added so that the step stays on one stream and kernel times are those of the kernel alone.  (Round 2's bits 65536 / 131072 belonged to the
# colour backward's own scatter, which round 3 replaced by train_bin_* kernels.)  Per-kernel figures from rocprofv3 --kernel-trace --stats.
# In-kernel section clocks and transaction counts: profiles/exp_train_sections.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ab in ${ABS:-524288 524544 524800 525056}; do
  rm -rf gpurun_out/ktt && mkdir -p gpurun_out/ktt
  NGF_ABLATE=$ab timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktt -o kt -- python profiles/workload.py train_R1 20 > /dev/null 2>&1
  echo "NGF_ABLATE=$ab: $(python profiles/summarize_rocpd.py $(find gpurun_out/ktt -name '*.db' | head -1) | grep -E "train_density_bwd|train_color_bwd")"
done
rm -rf gpurun_out/ktt
