# two-level block images (8- and 4-cell blocks): shipped library against build/exp/noskip, sparse / cluttered / full masks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NOSKIP=neural-gauge-fields_amd/csrc/build/exp/noskip/libngf_hip.so
for rep in 1 2; do
  for w in triplane_R1_bdc_S884ball triplane_R1_bdc_S884lattice triplane_R1_bdc_S884shell triplane_R0_bdc_S884lattice infoinv_R1__S884lattice infoinv_R1__S884ball triplane_R1_bdc_S884mask triplane_R2_bdc_S884mask triplane_R2_bdc_S884ball infoinv_R1__S884mask; do
    echo "skip    $(python profiles/workload.py $w 10 2>&1 | tail -1)"
    echo "no skip $(NGF_LIB=$NOSKIP python profiles/workload.py $w 10 2>&1 | tail -1)"
  done
done
