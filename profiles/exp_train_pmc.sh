#!/bin/bash
# PMC passes over the training-step kernels (run on the GPU box through gpurun): HBM/fabric traffic and busy fractions per kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_train; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" "TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python profiles/workload.py train_R1 4 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_train/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("ngf::", "")
        k = k.replace("void ", "")
        if not k.startswith(("train_", "xty", "colsum", "adam", "pack_plane")): continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':28s} {'fetch MB (x2)':>14s} {'write MB':>10s} {'MFMA busy':>10s} {'VALU busy':>10s} {'TA busy':>8s} {'L2 hit':>7s} {'L1 tags/cyc/CU':>15s} {'wait mem':>9s} {'wait issue':>11s} {'LDS busy':>9s} {'LDS confl':>10s}")
for k, c in sorted(vals.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    def pct(x): return f"{100*x:9.1f}%" if cyc else "       n/a"
    print(f"{k:28s} {2*m.get('FETCH_SIZE',0)*1024/1e6:14.1f} {m.get('WRITE_SIZE',0)*1024/1e6:10.1f} "
          f"{pct(m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(cyc*1024) if cyc else 0)} {pct(4*m.get('SQ_ACTIVE_INST_VALU',0)/(cyc*1024) if cyc else 0)} "
          f"{pct(m.get('TA_TA_BUSY_sum',0)/(cyc*256) if cyc else 0)[:8]} {100*m.get('TCC_HIT_sum',0)/max(m.get('TCC_HIT_sum',0)+m.get('TCC_MISS_sum',0),1):6.1f}% "
          f"{(m.get('TCP_TOTAL_CACHE_ACCESSES_sum',0)/(cyc*256) if cyc else 0):15.3f} {100*m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',0),1):8.1f}% {100*m.get('SQ_WAIT_INST_ANY',0)/max(m.get('SQ_WAVE_CYCLES',0),1):10.1f}% {pct(m.get('SQ_LDS_IDX_ACTIVE',0)/(cyc*256) if cyc else 0)[:9]} {100*m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',0),1):9.1f}%")
PY
rm -rf $OUT/p*/
