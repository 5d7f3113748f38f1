#!/bin/bash
# Does SQ_ACTIVE_INST_VALU count the issue cycles of MFMA instructions?  PMC passes over profiles/micro/mfma_valu_overlap (its kernel launches,
# in order: warm-up + timed launch of {fp32 MFMA only}, {fp32 VALU only}, {bf16 MFMA only}, ...): per-dispatch counters of the timed launches.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cs; rm -rf $OUT; mkdir -p $OUT
( cd profiles/micro && hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip 2>/dev/null )
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p1 -o p1 -- profiles/micro/mfma_valu_overlap > $OUT/run.log 2>&1
python - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/cs/p1/*counter_collection.csv"):
    rows += list(csv.DictReader(open(f)))
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
names = ["f32 MFMA only", "f32 VALU only", "bf16 MFMA only", "f32 MFMA + f32 MFMA", "VALU + VALU", "f32 MFMA + VALU", "bf16 MFMA + VALU", "bf16 + bf16", "f32 MFMA + bf16 MFMA",
         "1 wave: MFMA+2fma", "1 wave: MFMA+4fma", "1 wave: MFMA+6fma", "1 wave: bf16+2fma", "2 waves: MFMA+2fma", "2 waves: MFMA+6fma"]
ids = sorted(by)
timed = ids[1::2]          # every second dispatch is the timed launch
print(f"{'launch':24s} {'VALU busy':>10s} {'MFMA busy':>10s} {'coexec/MFMA':>12s}   (VALU busy = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024))")
for n, i in zip(names, timed):
    m = by[i]
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024
    if not cyc: continue
    print(f"{n:24s} {100*4*m.get('SQ_ACTIVE_INST_VALU',0)/cyc:9.1f}% {100*m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/cyc:9.1f}% {100*m.get('SQ_VALU_MFMA_COEXEC_CYCLES',0)/max(m.get('SQ_VALU_MFMA_BUSY_CYCLES',1),1):11.1f}%")
PY
rm -rf $OUT
