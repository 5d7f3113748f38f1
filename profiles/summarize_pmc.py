#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (profiles/collect.sh) for one kernel.

    python profiles/summarize_pmc.py <dir with p*/ passes> [out.json] [kernel-name substring]
Per-dispatch counter values are averaged over the dispatches of the named kernel whose grid is the largest seen (the
full-size launches; warm-up / stats launches of other sizes are ignored).  Units and corrections follow
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB, FETCH_SIZE is doubled (gfx950 tallies the
128-byte requests of wide reads at 64 bytes), SQ_*_CYCLES that count quad-cycles are used as ratios only, busy fractions are
per (active cycles x units).  The JSON carries the sha of the .so the run used, so bench.py can tell a stale summary.
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out_json = sys.argv[2] if len(sys.argv) > 2 else None
ksub = sys.argv[3] if len(sys.argv) > 3 else "ngf::render_kernel"

rows, traces = [], []
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    with open(f) as fh:
        rows += [r for r in csv.DictReader(fh) if ksub in r["Kernel_Name"]]
for f in sorted(glob.glob(os.path.join(root, "p*", "*kernel_trace.csv"))):
    with open(f) as fh:
        traces += [r for r in csv.DictReader(fh) if ksub in r["Kernel_Name"]]
if not rows:
    raise SystemExit(f"no dispatch of a kernel matching '{ksub}' under {root}")
gmax = max(int(r["Grid_Size"]) for r in rows)
rows = [r for r in rows if int(r["Grid_Size"]) == gmax]
# several instantiations may match (round 3: the workload's statistics launch runs the DBG = true kernel): keep the one launched most often
from collections import Counter
kmain = Counter(r["Kernel_Name"] for r in rows).most_common(1)[0][0]
rows = [r for r in rows if r["Kernel_Name"] == kmain]
traces = [r for r in traces if r["Kernel_Name"] == kmain]
gkey = next((k for k in ("Grid_Size", "Grid_Size_X") if traces and k in traces[0]), None)
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in traces if gkey is None or int(r[gkey]) == gmax]
vals = defaultdict(list)
for r in rows:
    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
r0 = rows[0]
kname = r0["Kernel_Name"]
print(f"# {root}: {kname[:150]}")
print(f"# grid {gmax} x wg {r0['Workgroup_Size']}, LDS {r0['LDS_Block_Size']} B, scratch {r0['Scratch_Size']} B/lane, VGPR {r0['VGPR_Count']} "
      f"AGPR {r0['Accum_VGPR_Count']} SGPR {r0['SGPR_Count']}; {len(durs)} dispatches in the kernel traces, mean {sum(durs) / max(len(durs), 1):.3f} ms (under PMC collection)")
m = {}
for k, v in sorted(vals.items()):
    m[k] = sum(v) / len(v)
    print(f"{k:34s} mean/dispatch {m[k]:.6g}   (n={len(v)})")
print()
ms = sum(durs) / max(len(durs), 1)
out = {"kernel": kname[:200], "kernel_ms_under_pmc": ms, "vgpr": int(r0["VGPR_Count"]), "agpr": int(r0["Accum_VGPR_Count"]),
       "lds_bytes": int(r0["LDS_Block_Size"]), "scratch_bytes_per_lane": int(r0["Scratch_Size"])}
if "FETCH_SIZE" in m:
    fs = m["FETCH_SIZE"] * 1024
    print(f"FETCH_SIZE  = {fs / 1e6:.1f} MB/dispatch as reported; x2 gfx950 wide-read correction (MI355X_MICROARCH.md HBM section) = {2 * fs / 1e6:.1f} MB")
    out["fetch_bytes_corrected"] = 2 * fs
    out["write_bytes"] = m.get("WRITE_SIZE", 0) * 1024
    out["hbm_traffic_bytes_per_launch"] = out["fetch_bytes_corrected"] + out["write_bytes"]
    print(f"fabric traffic = {out['hbm_traffic_bytes_per_launch'] / 1e9:.3f} GB/dispatch = {out['hbm_traffic_bytes_per_launch'] / (ms * 1e-3) / 1e12:.3f} TB/s "
          f"({100 * out['hbm_traffic_bytes_per_launch'] / (ms * 1e-3) / 8e12:.1f} % of the 8 TB/s HBM peak)")
if "WRITE_SIZE" in m:
    print(f"WRITE_SIZE  = {m['WRITE_SIZE'] * 1024 / 1e6:.1f} MB/dispatch (uncalibrated)")
if "TCC_HIT_sum" in m:
    out["l2_hit_frac"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    print(f"L2 hit rate = {100 * out['l2_hit_frac']:.1f} %")
if "GRBM_GUI_ACTIVE" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
    print(f"active cycles per XCD = {cyc:.4g}")
    if durs:
        out["effective_clock_ghz"] = cyc / (ms * 1e-3) / 1e9
        print(f"effective clock ~ {out['effective_clock_ghz']:.2f} GHz")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        out["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
        print(f"MFMA busy   = {100 * out['mfma_busy_frac']:.1f} % of (active cycles x 1024 SIMDs)")
    if "TA_TA_BUSY_sum" in m:
        out["ta_busy_frac"] = m["TA_TA_BUSY_sum"] / (cyc * 256)
        print(f"TA busy     = {100 * out['ta_busy_frac']:.1f} % of (active cycles x 256 CUs)")
    if "SQ_ACTIVE_INST_VALU" in m:
        out["valu_busy_frac"] = 4 * m["SQ_ACTIVE_INST_VALU"] / (cyc * 1024)
        print(f"VALU busy   = {100 * out['valu_busy_frac']:.1f} % (SQ_ACTIVE_INST_VALU quad-cycles x4 / SIMD cycles)")
    if "SQ_LDS_IDX_ACTIVE" in m:
        out["lds_busy_frac"] = m["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
        out["lds_bank_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m["SQ_LDS_IDX_ACTIVE"], 1.0)
        print(f"LDS busy    = {100 * out['lds_busy_frac']:.1f} % of (active cycles x 256 CUs); bank conflicts {100 * out['lds_bank_conflict_frac']:.1f} % of LDS-active cycles")
    if "SQ_WAVE_CYCLES" in m:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in m:
                out[k.lower() + "_frac"] = m[k] / m["SQ_WAVE_CYCLES"]
                print(f"{k:20s}= {100 * m[k] / m['SQ_WAVE_CYCLES']:.1f} % of wave cycles")
if "TCP_TOTAL_CACHE_ACCESSES_sum" in m and "TCP_TCC_READ_REQ_sum" in m:
    out["l1_hit_frac"] = 1 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"]
    out["tcp_accesses_per_launch"] = m["TCP_TOTAL_CACHE_ACCESSES_sum"]
    out["l2_read_requests_per_launch"] = m["TCP_TCC_READ_REQ_sum"]
    # L1 -> L2 read requests are 64-byte requests on gfx9-family TCPs (128-byte lines are fetched as two): bytes the L2 served
    out["l2_read_bytes_per_launch"] = m["TCP_TCC_READ_REQ_sum"] * 64.0
    print(f"L1 hit rate ~ {100 * out['l1_hit_frac']:.1f} % (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES); L1 tag accesses {m['TCP_TOTAL_CACHE_ACCESSES_sum']:.4g} "
          f"= {m['TCP_TOTAL_CACHE_ACCESSES_sum'] / (ms * 1e-3) / 1e9:.1f} G/s")
    print(f"L2 read traffic ~ {out['l2_read_bytes_per_launch'] / 1e9:.2f} GB/dispatch (x64 B per TCP->TCC request) = {out['l2_read_bytes_per_launch'] / (ms * 1e-3) / 1e12:.2f} TB/s "
          f"({100 * out['l2_read_bytes_per_launch'] / (ms * 1e-3) / 34.5e12:.1f} % of the ~34.5 TB/s L2 peak)")
if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
    out["mfma_flops_per_dispatch"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
    print(f"MFMA f32 flops/dispatch = {out['mfma_flops_per_dispatch']:.4g} -> {out['mfma_flops_per_dispatch'] / (ms * 1e-3) / 1e12:.1f} TFLOP/s "
          f"= {100 * out['mfma_flops_per_dispatch'] / (ms * 1e-3) / 157.3e12:.1f} % of the 157.3 TFLOP/s fp32 matrix peak (at the PMC-run duration)")
if m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) > 0:
    out["mfma_bf16_flops_per_dispatch"] = m["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
    print(f"MFMA bf16 flops/dispatch = {out['mfma_bf16_flops_per_dispatch']:.4g} -> {out['mfma_bf16_flops_per_dispatch'] / (ms * 1e-3) / 1e12:.1f} TFLOP/s "
          f"= {100 * out['mfma_bf16_flops_per_dispatch'] / (ms * 1e-3) / 2500e12:.1f} % of the ~2500 TFLOP/s dense bf16 matrix peak")
if "GRBM_GUI_ACTIVE" in m and "valu_busy_frac" in out and "mfma_busy_frac" in out:
    # SQ_ACTIVE_INST_VALU also counts the ISSUE cycles of MFMA instructions (profiles/exp_counter_semantics.sh on the micro-benchmark:
    # a wave that only issues v_mfma_f32_16x16x4_f32 shows 12.4 % "VALU busy" next to 99.6 % MFMA busy = 4 issue cycles of 33;
    # bf16 16x16x32: 24.5 % next to 97.8 % = 4 of 16).  Issue cycles per flop are the same for the two tile shapes of a type:
    # 4 cycles per 2048 fp32 flops (8 per 4096), 4 per 16384 bf16 flops (8 per 32768).
    issue = (out.get("mfma_flops_per_dispatch", 0.0) / 2048.0 + out.get("mfma_bf16_flops_per_dispatch", 0.0) / 16384.0) * 4.0
    out["mfma_issue_frac"] = issue / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
    out["valu_busy_excl_mfma_issue_frac"] = max(out["valu_busy_frac"] - out["mfma_issue_frac"], 0.0)
    out["simd_busy_frac"] = out["mfma_busy_frac"] + out["valu_busy_excl_mfma_issue_frac"]
    print(f"MFMA issue  = {100 * out['mfma_issue_frac']:.1f} % of the SIMD cycles are inside VALU busy as well -> VALU without it {100 * out['valu_busy_excl_mfma_issue_frac']:.1f} %, "
          f"SIMD busy (MFMA + other VALU) = {100 * out['simd_busy_frac']:.1f} %")
if "SQ_VALU_MFMA_COEXEC_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
    out["mfma_valu_coexec_frac_of_mfma_busy"] = m["SQ_VALU_MFMA_COEXEC_CYCLES"] / max(m["SQ_VALU_MFMA_BUSY_CYCLES"], 1.0)
    print(f"VALU / MFMA co-execution = {100 * out['mfma_valu_coexec_frac_of_mfma_busy']:.1f} % of the MFMA-busy cycles (SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES)")
if "SQ_INSTS_VALU" in m:
    out["valu_insts_per_launch"] = m["SQ_INSTS_VALU"]          # wave-level VALU instructions, MFMA included
    print(f"VALU instructions / dispatch = {m['SQ_INSTS_VALU']:.4g} (wave level, MFMA included: {out.get('mfma_flops_per_dispatch', 0.0) / 2048.0:.4g} fp32 MFMA 16x16x4-equivalents)")
if "SQ_INSTS_VMEM_RD" in m:
    out["vmem_read_insts_per_launch"] = m["SQ_INSTS_VMEM_RD"]
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neural-gauge-fields_amd", "csrc", "libngf_hip.so")
if os.path.exists(so):
    out["so_sha16"] = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
    print(f"libngf_hip.so sha256[:16] = {out['so_sha16']}")
if out_json:
    json.dump(out, open(out_json, "w"), indent=1)
