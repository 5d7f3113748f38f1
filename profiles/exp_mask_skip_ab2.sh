# A/B of the empty-space skipping on cluttered occupancy (thin-wall lattice, thin shell): shipped library against build/exp/noskip
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NOSKIP=neural-gauge-fields_amd/csrc/build/exp/noskip/libngf_hip.so
for rep in 1 2; do
  for w in triplane_R1_bdc_S884lattice triplane_R1_bdc_S884shell triplane_R0_bdc_S884lattice infoinv_R1__S884lattice; do
    echo "skip    $(python profiles/workload.py $w 10 2>&1 | tail -1)"
    echo "no skip $(NGF_LIB=$NOSKIP python profiles/workload.py $w 10 2>&1 | tail -1)"
  done
done
