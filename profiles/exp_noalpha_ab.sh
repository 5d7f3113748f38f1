# A/B: the march without the alpha-mask test compiled in (build/exp/noalpha, -DNGF_EXP_NO_ALPHA_MASK=1) against the shipped library, frames WITHOUT a mask
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ALT=neural-gauge-fields_amd/csrc/build/exp/noalpha/libngf_hip.so
for rep in 1 2 3; do
  for w in triplane_R1_bdc triplane_R2_bdc infoinv_R1 triplane_R1_bd; do
    echo "shipped  $(python profiles/workload.py $w 20 2>&1 | tail -1)"
    echo "no alpha $(NGF_LIB=$ALT python profiles/workload.py $w 20 2>&1 | tail -1)"
  done
done
