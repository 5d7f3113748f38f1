#!/usr/bin/env python3
"""One rank's pipeline of `bench.py --gpus 8` on ONE GPU (as exp_pipeline_gap.py: world-1 RCCL group, 80 000-ray shard launches): consecutive frames on ONE
render stream against frames alternating between TWO render streams.  On one stream frame k + 1 starts when the last wave of frame k has ended: the
launch's tail (wave slots idle at the ends: ~45 us, profiles/r06_timeline.txt) and the launch gap are paid per frame.  On two streams the next frame's workgroups
take the CUs the previous frame has already left (a level-3 workgroup needs a whole CU: 12 waves x 168 registers), so the tail is filled with the next
frame's head.  Same pixels (checked).
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29583 RANK=0 WORLD_SIZE=1 python profiles/exp_two_streams.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import ngf_amd  # noqa: F401
from ngf_amd import cases, dist as ndist, synth

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29583")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
WORLD = int(os.environ.get("SHARDS", "8"))
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device="cuda", bake=True, bake_color=True)
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda().view(800, 800, 6)
rows = ndist.interleaved_rows(800, WORLD, 0, 10)
rays = torch.cat([frame[a:b] for a, b in rows]).reshape(-1, 6).contiguous()
per = rays.shape[0]
DEPTH = int(os.environ.get("DEPTH", ndist.PIPELINE_DEPTH))          # as bench.py: PipelinedGather(depth=ndist.PIPELINE_DEPTH)
pipe = ndist.PipelinedGather(per, 1, dev, depth=DEPTH)
side = torch.cuda.Stream(dev)
rstreams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
out = [(torch.empty((per, 3), device=dev), torch.empty((per,), device=dev)) for _ in range(DEPTH)]
ref = f(rays, N_samples=192, white_bg=True, iteration=30001, row_width=800)
ref = (ref["rgb_map"].clone(), ref["depth_map"].clone())


def run(steps, nstreams, exchange=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        st = rstreams[k % nstreams] if nstreams > 0 else torch.cuda.current_stream()
        with torch.cuda.stream(st):
            o = pipe.buffers(k) if exchange else out[k % DEPTH]
            f(rays, N_samples=192, white_bg=True, out=o, iteration=30001, row_width=800)
            if exchange:
                pipe.submit(k)
        if exchange and k > 0:
            pipe.frame_in_image_order(k - 1, 800 // WORLD, 800, 10, out=out[(k - 1) % DEPTH], stream=side)
    if exchange:
        pipe.frame_in_image_order(steps - 1, 800 // WORLD, 800, 10, out=out[(steps - 1) % DEPTH], stream=side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print(f"{per} rays per step (rank 0 of {WORLD}), level 3; ms per step, median of 5 runs of 100 steps")
if os.environ.get("TAILS"):
    # does a launch whose tail is filled by the next frame want fewer narrow tiles?  (knob tail: sixteenths of a tile per resident wave and width; 16 = default)
    from ngf_amd import _lib
    for tail in [int(x) for x in os.environ["TAILS"].split(",")]:
        with _lib.knobs(tail=tail):
            run(10, 1); run(10, 2)
            a, b = [], []
            for rep in range(5):
                a.append(run(100, 1)); b.append(run(100, 2))
        print(f"tail = {tail:3d}: one side stream {np.median(a):.4f}   two alternating streams {np.median(b):.4f}")
for exchange in (False, True):
    for ns in (0, 1, 2):
        run(10, ns, exchange)
    res = {0: [], 1: [], 2: []}
    for rep in range(5):
        for ns in (0, 1, 2):
            res[ns].append(run(100, ns, exchange))
    same = all(torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]) for o in out) if exchange else True
    print(f"{'render + exchange + reorder' if exchange else 'render launches only       '}: current stream {np.median(res[0]):.4f}   one side stream {np.median(res[1]):.4f}   two alternating streams {np.median(res[2]):.4f}"
          + (f"   gathered frames bit-identical: {same}" if exchange else ""))
dist.destroy_process_group()
