cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=neural-gauge-fields_amd/csrc
for rep in 1 2; do
for v in base ilp memcl; do
  if [ $v = base ]; then L=$B/libngf_hip.so; else L=$B/build/exp/$v/libngf_hip.so; fi
  for wl in triplane_R1_bdc infoinv_R1 triplane_R2_bdc train_R1; do
    echo -n "$v $wl: "; NGF_LIB=$L timeout 200 python profiles/workload.py $wl 6 2>/dev/null | grep -v amdgpu.ids | tail -1
  done
done; done
