#!/bin/bash
# Experiment builds behind profiles/r03_determinism.txt (run in the build container; the .so files travel with gpurun, build/ is git-ignored):
#   packed       the compiler's packed code for the positional-factor chain of the InfoInv colour passes (pe_octave, ngf_infoinv.hpp) -- round 2's code
#   packed_nops  the same + 32 idle issue slots around every bf16 MFMA pair: the nondeterminism goes from 1 launch in 50 000 to 1 in 5
#   nops         the shipped (un-packed) code with the idle slots: stays bit-stable
#   dump         packed_nops + a per-lane dump of every colour pass (profiles/exp_determinism_dump.py)
# Use:  NGF_LIB=neural-gauge-fields_amd/csrc/build/exp/<name>/libngf_hip.so python profiles/exp_determinism_fast.py 300000 0 -1 -1 infoinv_r1_on/split
cd "$(dirname "$0")/../neural-gauge-fields_amd/csrc"
make libngf_hip.so
make exp NAME=packed DEFS="-DNGF_EXP_PACKED_PE=1" &
make exp NAME=packed_nops DEFS="-DNGF_EXP_PACKED_PE=1 -DNGF_EXP_NOPS=1" &
wait
make exp NAME=nops DEFS="-DNGF_EXP_NOPS=1" &
make exp NAME=dump DEFS="-DNGF_EXP_PACKED_PE=1 -DNGF_EXP_NOPS=1 -DNGF_EXP_DUMP=1" &
wait
ls -la build/exp/*/libngf_hip.so
