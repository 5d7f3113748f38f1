#!/usr/bin/env python3
"""Does the exchange of frame k run BESIDE the render of frame k + 1, or only between two render launches?  The level-3 render kernel keeps 12 waves x 168
registers on every CU: 504 of a SIMD's 512 registers, nothing else fits while its workgroups are resident.  One rank's pipeline of `bench.py --gpus 8` on ONE
GPU (world-1 RCCL group: the all-gather is a local copy; a 10 MB device copy stands in for the eight ranks' reorder): 80 000-ray shard launches, the step's
wall clock against the launch's own duration, with the render on all CUs and with a few CUs left free (knob "grid").
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 RANK=0 WORLD_SIZE=1 python profiles/exp_pipeline_gap.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import ngf_amd  # noqa: F401
from ngf_amd import _lib, cases, dist as ndist, synth

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device="cuda", bake=True, bake_color=True)
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda().view(800, 800, 6)
rows = ndist.interleaved_rows(800, 8, 0, 10)
rays = torch.cat([frame[a:b] for a, b in rows]).reshape(-1, 6).contiguous()
per = rays.shape[0]
pipe = ndist.PipelinedGather(per, 1, dev)
side = torch.cuda.Stream(dev)
out = [(torch.empty((per, 3), device=dev), torch.empty((per,), device=dev)) for _ in range(2)]
big_src, big_dst = torch.empty(2_560_000, device=dev), torch.empty(2_560_000, device=dev)      # 10 MB: the reorder of an eight-rank frame


def run(steps, extra_copy):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        o = pipe.buffers(k)
        ev[k][0].record()
        f(rays, N_samples=192, white_bg=True, out=o, iteration=30001)
        ev[k][1].record()
        pipe.submit(k)
        if k > 0:
            pipe.frame_in_image_order(k - 1, 100, 800, 10, out=out[(k - 1) % 2], stream=side)
            if extra_copy:
                with torch.cuda.stream(side):
                    big_dst.copy_(big_src)
    pipe.frame_in_image_order(steps - 1, 100, 800, 10, out=out[(steps - 1) % 2], stream=side)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps * 1e3
    return el, float(np.median([a.elapsed_time(b) for a, b in ev]))


def plain(steps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    o = pipe.buffers(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record(); f(rays, N_samples=192, white_bg=True, out=o, iteration=30001); ev[k][1].record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(np.median([a.elapsed_time(b) for a, b in ev]))


print(f"{per} rays per step (rank 0 of 8), level 3")
for grid in (-1, 252, 248, 240, 224):
    with _lib.knobs(grid=grid):
        plain(10); run(10, True)
        p = plain(60)
        a = run(60, False)
        b = run(60, True)
    print(f"render on {'all' if grid < 0 else grid} CUs: launches alone {p[0]:.4f} ms/step (launch {p[1]:.4f}); + exchange + 1.3 MB reorder {a[0]:.4f} (launch {a[1]:.4f}); "
          f"+ exchange + 10 MB reorder {b[0]:.4f} ms/step (launch {b[1]:.4f})")
dist.destroy_process_group()
