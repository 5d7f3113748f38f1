#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a small text summary.

    python profiles/summarize_rocpd.py gpurun_out/prof_xx/xx_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
print("# kernel-trace summary (rocprofv3 --kernel-trace --stats), durations in microseconds")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'min_us':>12} {'max_us':>12} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7}  name")
rows = cur.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration),
                      max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size)
                      from kernels group by name order by sum(duration) desc""").fetchall()
for name, calls, tot, avg, mn, mx, vg, ag, sg, lds, scr in rows:
    print(f"{calls:>6} {tot/1e3:>12.1f} {avg/1e3:>12.1f} {mn/1e3:>12.1f} {mx/1e3:>12.1f} {vg:>5} {ag:>5} {sg:>5} {lds:>7} {scr:>7}  {name[:110]}")
