#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (profiles/collect_pmc.sh) for the render kernel.

    python profiles/summarize_pmc.py gpurun_out/pmc_<tag> > profiles/<tag>_pmc.txt
Per-dispatch counter values are averaged over the full-frame launches of ngf::render_kernel.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
vals = defaultdict(list)
durs = []
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "render_kernel" not in r["Kernel_Name"]:
                continue
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(root, "p*", "*kernel_trace.csv"))):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "render_kernel" in r["Kernel_Name"]:
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print(f"# {root}: ngf::render_kernel, {len(durs)} dispatches in kernel traces, mean {sum(durs)/max(len(durs),1):.3f} ms (under PMC collection)")
m = {}
for k, v in sorted(vals.items()):
    m[k] = sum(v) / len(v)
    print(f"{k:34s} mean/dispatch {m[k]:.6g}   (n={len(v)})")
print()
if "FETCH_SIZE" in m:
    fs = m["FETCH_SIZE"] * 1024
    print(f"FETCH_SIZE  = {fs/1e6:.1f} MB/dispatch as reported; x2 gfx950 wide-read correction (MI355X_MICROARCH.md HBM section) = {2*fs/1e6:.1f} MB")
if "WRITE_SIZE" in m:
    print(f"WRITE_SIZE  = {m['WRITE_SIZE']*1024/1e6:.1f} MB/dispatch (uncalibrated)")
if "TCC_HIT_sum" in m:
    print(f"L2 hit rate = {100*m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum']):.1f} %")
out = {"kernel_ms_under_pmc": sum(durs) / max(len(durs), 1)}
if "GRBM_GUI_ACTIVE" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
    print(f"active cycles per XCD = {cyc:.4g}")
    if durs:
        print(f"effective clock ~ {cyc/(sum(durs)/len(durs)*1e-3)/1e9:.2f} GHz")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        out["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
        print(f"MFMA busy   = {100*out['mfma_busy_frac']:.1f} % of (active cycles x 1024 SIMDs)")
    if "TA_TA_BUSY_sum" in m:
        out["ta_busy_frac"] = m["TA_TA_BUSY_sum"] / (cyc * 256)
        print(f"TA busy     = {100*out['ta_busy_frac']:.1f} % of (active cycles x 256 CUs)")
    if "SQ_ACTIVE_INST_VALU" in m:
        out["valu_busy_frac"] = 4 * m["SQ_ACTIVE_INST_VALU"] / (cyc * 1024)
        print(f"VALU busy   = {100*out['valu_busy_frac']:.1f} % (SQ_ACTIVE_INST_VALU quad-cycles x4 / SIMD cycles)")
    if "SQ_WAVE_CYCLES" in m:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in m:
                print(f"{k:20s}= {100*m[k]/m['SQ_WAVE_CYCLES']:.1f} % of wave cycles")
if "TCP_TOTAL_CACHE_ACCESSES_sum" in m and "TCP_TCC_READ_REQ_sum" in m:
    out["l1_hit_frac"] = 1 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"]
    print(f"L1 hit rate ~ {100*out['l1_hit_frac']:.1f} % (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)")
if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
    out["mfma_flops_per_dispatch"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
    print(f"MFMA f32 flops/dispatch = {m['SQ_INSTS_VALU_MFMA_MOPS_F32']*512:.4g}")
if "FETCH_SIZE" in m:
    out["fetch_bytes_corrected"] = 2 * m["FETCH_SIZE"] * 1024
    out["write_bytes"] = m.get("WRITE_SIZE", 0) * 1024
    out["hbm_traffic_bytes_per_launch"] = out["fetch_bytes_corrected"] + out["write_bytes"]
if "TCC_HIT_sum" in m:
    out["l2_hit_frac"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
if len(sys.argv) > 2:
    import json
    json.dump(out, open(sys.argv[2], "w"), indent=1)
