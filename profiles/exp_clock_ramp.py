#!/usr/bin/env python3
"""Per-launch duration of the first 400 launches of a process (80 000-ray shard, 320 000-ray shard, full frame back to back): where are the slow launches that
make the first timed block of a script 4-8 % slower than its later ones (profiles/exp_addr.py: not the addresses)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ngf_amd  # noqa: F401
from ngf_amd import cases, dist, synth
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device="cuda", bake=True, bake_color=True)
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda().view(800, 800, 6)
def shard(world):
    return torch.cat([frame[a:b] for a, b in dist.interleaved_rows(800, world, 0, 10)]).reshape(-1, 6).contiguous()
for label, rays, n in (("320000 rays", shard(2), 400), ("640000 rays", shard(1), 200), ("80000 rays", shard(8), 800), ("320000 rays again", shard(2), 100)):
    out = (torch.empty((rays.shape[0], 3), device="cuda"), torch.empty((rays.shape[0],), device="cuda"))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    idle = float(os.environ.get("IDLE_S", "0"))
    if idle: time.sleep(idle)
    ev[0].record()
    for k in range(n):
        f(rays, N_samples=192, white_bg=True, iteration=30001, out=out, row_width=800); ev[k + 1].record()
    torch.cuda.synchronize()
    ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(n)])
    t = np.cumsum(ms)
    med = np.median(ms[n // 2:])
    slow = ms > 1.02 * med
    print(f"{label}: steady median {med:.4f} ms; launches > 1.02 x median: {int(slow.sum())} of {n}; by 50-launch block (mean / median): "
          + " ".join(f"{ms[i:i + 50].mean() / med:.3f}" for i in range(0, n, 50)))
    if slow.any():
        idx = np.nonzero(slow)[0]
        print(f"    slow launches at indices {idx[:12].tolist()}{' ...' if len(idx) > 12 else ''} (t = {t[idx[0]]:.0f} .. {t[idx[-1]]:.0f} ms after the block's start), worst {ms.max() / med:.3f} x")
