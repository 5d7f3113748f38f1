#!/usr/bin/env python3
"""VGPR / AGPR split / scratch / LDS of every kernel in a hipcc -save-temps assembly file (the .amdhsa_kernel descriptors) and, with a
kernel-name substring as second argument, that kernel's scratch instructions.
    python profiles/kernel_resources.py neural-gauge-fields_amd/csrc/build/asm/ngf_field-hip-amdgcn-amd-amdhsa-gfx950.s [substring]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for n in re.findall(r"\.amdhsa_kernel (\S+)", s):
    i = s.index(".amdhsa_kernel " + n)
    blk = s[i:i + 6000]
    v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", blk).group(1)
    a = re.search(r"\.amdhsa_accum_offset (\d+)", blk)
    sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", blk).group(1)
    dn = n
    for x, y in (("_ZN3ngf13render_kernelINS_", "render<"), ("EEvNS_10RenderArgsE", ">"), ("_ZN3ngf", "")):
        dn = dn.replace(x, y)
    if want in n:
        print(f"{dn[:96]:96s} vgpr {v:>4s} accum_offset {a.group(1) if a else '-':>4s} scratch {sc:>4s} B/lane")
        if want:
            j = s.index("\n" + n + ":")
            body = s[j:s.index("s_endpgm", j)]
            for k, l in enumerate(body.split("\n")):
                if "scratch_" in l:
                    print(f"      +{k}: {l.strip()[:110]}")
