import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
sys.argv = [sys.argv[0]]
os.environ["WORLDS"] = ""
exec(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "profiles/exp_rank_shards.py")).read().split("full = timed(")[0])
full = timed(frame.reshape(-1, 6).contiguous())
print("full", full)
for rank in (0, 1):
    rows = dist.interleaved_rows(800, 2, rank, 10)
    rays = torch.cat([frame[a:b] for a, b in rows]).reshape(-1, 6).contiguous()
    a = timed(rays); b = timed(rays); c = rays.clone(); d = timed(c); e = timed(rays)
    print(f"rank {rank}: rays @{rays.data_ptr():#x} {a:.3f} again {b:.3f}; clone @{c.data_ptr():#x} {d:.3f}; original again {e:.3f}")
    del c
