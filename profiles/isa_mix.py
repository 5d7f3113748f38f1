#!/usr/bin/env python3
"""Static instruction mix of one kernel of a hipcc -save-temps assembly file, per basic block and in total:
    make -C neural-gauge-fields_amd/csrc asm            # writes build/asm/*.s
    python profiles/isa_mix.py neural-gauge-fields_amd/csrc/build/asm/ngf_field-hip-amdgcn-amd-amdhsa-gfx950.s 'render_kernelINS_14TriPlanePolicyILb1ELb0ELi12ELi1ELb0EEELb1E' [min_block]
Counts VALU (non-MFMA v_*), MFMA, SALU, LDS (ds_*), VMEM (global/buffer/flat/scratch) instructions; v_readlane/v_writelane (SGPR spills
parked in VGPR lanes) are listed on their own because they cost VALU issue slots."""
import re
import sys


def mix(path, needle, min_block=15):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(needle) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], None
    for i in range(start, end + 1):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m or cur is None:
            cur = [m.group(1) if m else "entry", i - start, {}]
            blocks.append(cur)
            if m:
                continue
        s = l.strip()
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("v_mfma"): k = "mfma"
        elif op.startswith(("v_readlane", "v_writelane")): k = "lane"
        elif op.startswith("v_"): k = "valu"
        elif op.startswith("s_"): k = "salu"
        elif op.startswith("ds_"): k = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = "vmem"
        else: k = "other"
        cur[2][k] = cur[2].get(k, 0) + 1
    tot = {}
    for b in blocks:
        for k, v in b[2].items():
            tot[k] = tot.get(k, 0) + v
        if sum(b[2].values()) >= min_block:
            print(f"{b[0]:>14} @{b[1]:5d} {b[2]}")
    print("TOTAL", lines[start].split(":")[0][:120], tot)
    for l in lines[end:end + 60]:
        if re.search(r"(NumVgprs|NumAgprs|ScratchSize|NumSgprs|Occupancy|LDSByteSize)", l):
            print("   ", l.strip())


if __name__ == "__main__":
    mix(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 15)
