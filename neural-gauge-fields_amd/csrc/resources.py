#!/usr/bin/env python3
"""make resources | python resources.py  -- one line per kernel: VGPRs, spills, scratch, occupancy."""
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = k.replace("_ZN3ngf13render_kernelINS_", "render<").replace("EvNS_10RenderArgsE", ">")
    print(f"{name[:70]:70s} VGPR {v.get('VGPRs',0):4d} spill {v.get('VGPRs Spill',0):4d}  SGPR spill {v.get('SGPRs Spill',0):4d}  scratch {v.get('ScratchSize [bytes/lane]',0):5d}  occ {v.get('Occupancy [waves/SIMD]',0)}")
