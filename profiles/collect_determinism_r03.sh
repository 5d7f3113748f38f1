#!/bin/bash
# profiles/r03_determinism.txt: the four experiment builds (profiles/exp_determinism_builds.sh) and the shipped library on the
# InfoInv NGF_F_SPLIT_BF16 case, 300 000 launches each, then the per-lane dump of three bad launches of the amplified build.
out=gpurun_out/r03_determinism.txt
{
echo "# InfoInv NGF_F_SPLIT_BF16 (infoinv_r1_on, 128 rays): launches whose frame differs bitwise from the first launch of the same handle"
echo "# builds: profiles/exp_determinism_builds.sh;  script: profiles/exp_determinism_fast.py <launches> 0 -1 -1 infoinv_r1_on/split"
for v in packed packed_nops nops; do
  NGF_LIB=neural-gauge-fields_amd/csrc/build/exp/$v/libngf_hip.so python profiles/exp_determinism_fast.py 300000 0 -1 -1 infoinv_r1_on/split 2>&1 | grep -v "amdgpu.ids\|     ray \|render differs"
done
echo "# shipped library (un-packed pe_octave), every case of the script:"
python profiles/exp_determinism_fast.py 300000 0 2>&1 | grep -v "amdgpu.ids\|     ray \|render differs"
echo "# per-lane dump of bad launches of the amplified build (packed + idle slots): which quantities differ between a good and a bad launch"
NGF_LIB=neural-gauge-fields_amd/csrc/build/exp/dump/libngf_hip.so python profiles/exp_determinism_dump.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 | head -40
} > $out 2>&1
tail -30 $out
