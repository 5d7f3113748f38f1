# A/B of the empty-space skipping (round 6): the shipped library against build/exp/noskip (make -C neural-gauge-fields_amd/csrc exp NAME=noskip DEFS=-DNGF_EXP_NO_MASK_SKIP=1),
# alternating runs inside one gpurun call; 10 frames each, median.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NOSKIP=neural-gauge-fields_amd/csrc/build/exp/noskip/libngf_hip.so
for rep in 1 2; do
  for w in triplane_R1_bdc_S884mask triplane_R1_bdc_S884ball triplane_R2_bdc_S884mask triplane_R2_bdc_S884ball triplane_R1_bdcs_S884ball infoinv_R1_split_S884ball infoinv_R1__S884ball infoinv_R1__S884mask triplane_R1_bdc infoinv_R1; do
    echo "skip    $(python profiles/workload.py $w 10 2>&1 | tail -1)"
    echo "no skip $(NGF_LIB=$NOSKIP python profiles/workload.py $w 10 2>&1 | tail -1)"
  done
done
