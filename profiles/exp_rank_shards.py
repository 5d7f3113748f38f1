#!/usr/bin/env python3
"""What eight ranks would each launch: the 800x800 R1 frame dealt out in 10-row blocks round robin (ngf_amd.dist.interleaved_rows, what bench.py --gpus 8
renders per rank), every rank's 80 000-ray shard timed on ONE MI355X next to the full frame.  kernel-only ceiling of N ranks = frame / slowest shard.
LEVEL=3 (module default) | 2.  Product-side imports only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ngf_amd  # noqa: F401
from ngf_amd import _lib, cases, dist, synth

if os.environ.get("NGF_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])

LEVEL = int(os.environ.get("LEVEL", "3"))
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device="cuda", bake=True, bake_color=LEVEL >= 3)
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda().view(800, 800, 6)


def timed(rays, rep=20):
    """Median launch of `rep`, outputs owned by the caller (torch's caching allocator may map a new segment for a new output size in the middle of a timed
    block: one 60 ms stall, then 12 launches on the clock ramp -- profiles/r06_clock_ramp.txt) and the clocks up before the first event."""
    import time
    out = (torch.empty((rays.shape[0], 3), device="cuda"), torch.empty((rays.shape[0],), device="cuda"))
    kw = dict(N_samples=192, white_bg=True, iteration=30001, out=out, row_width=800)          # as bench.py renders a rank's rows
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:
        for _ in range(3):
            f(rays, **kw)
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rep)]
    for a, b in ev:
        a.record(); f(rays, **kw); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


full = timed(frame.reshape(-1, 6).contiguous())
print(f"level {LEVEL}: full frame {full:.3f} ms = {640000 / full / 1e3:.1f} Mray/s")
for world in [int(v) for v in os.environ.get("WORLDS", "2 4 8").split()]:
    ms = []
    for rank in range(world):
        rows = dist.interleaved_rows(800, world, rank, 10)
        rays = torch.cat([frame[a:b] for a, b in rows]).reshape(-1, 6).contiguous()
        ms.append(timed(rays))
    print(f"  {world} ranks: shard launches {min(ms):.3f} .. {max(ms):.3f} ms ({640000 // world} rays each); frame / slowest shard = {full / max(ms):.2f}x "
          f"(sum of the shards {sum(ms):.3f} ms = {sum(ms) / full:.3f} of the frame)")
