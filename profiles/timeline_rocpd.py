#!/usr/bin/env python3
"""Timeline of the LAST training iteration in a rocprofv3 kernel trace (rocpd sqlite): start, end, duration, queue and name of every
dispatch between the last two train_density_kernel launches -- where the step's streams overlap and where they wait.
    python profiles/timeline_rocpd.py <results.db>"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f"select name, start, end, {qcol if qcol else '0'} from kernels order by start"
rows = cur.execute(sel).fetchall()
marks = [k for k, r in enumerate(rows) if "train_density_kernel" in r[0]]
if len(marks) < 3:
    sys.exit("fewer than three iterations in the trace")
a, b = marks[-3], marks[-2]          # a full iteration away from the end of the run
while a > 0 and "train_project_density" in rows[a - 1][0] or "fill" in rows[a - 1][0].lower() or "train_fold" in rows[a - 1][0]:
    a -= 1
t0 = rows[a][1]
print(f"{'start_us':>9} {'end_us':>9} {'dur_us':>8} {'queue':>6}  name")
for name, s, e, q in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {q:>6}  {name[:90]}")
