#!/usr/bin/env python3
"""Is a shard slow because it is SHORT or because of what it holds?  The 80 000 rays of rows 350-450 rendered 1x, 2x, 4x, 8x in ONE launch
(the same rays repeated), with the tile width forced to 4 and 8: if the 8x launch takes 8 x the full-frame rate the loss is launch-size
(ramp, tail, phase correlation of the waves); if it takes 8 x the 1x time it is the rays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ngf_amd  # noqa
from ngf_amd import _lib, synth
from ngf_amd.cases import big_case, field_for_case
g, params, step = big_case("triplane", "R1")
f = field_for_case(g, params, None, device="cuda", bake=True)
shard = torch.from_numpy(synth.lookat_rays(800, 800, rows=(350, 450))).cuda()
full = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
def t(rays, **knob):
    with _lib.knobs(**knob):
        for _ in range(3): f(rays, N_samples=192, iteration=30001)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); f(rays, N_samples=192, iteration=30001); b.record()
        torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for tw in (4, 8):
    for rep in (1, 2, 4, 8):
        rays = shard.repeat(rep, 1).contiguous()
        ms = t(rays, tile_w=tw)
        print(f"tile_w {tw}: shard x{rep} = {rays.shape[0]} rays: {ms:.4f} ms = {ms / rep:.4f} ms per shard ({rays.shape[0] / ms / 1e3:.1f} Mray/s)")
    ms = t(full, tile_w=tw)
    print(f"tile_w {tw}: full frame: {ms:.4f} ms ({640 / ms:.1f} Mray/s)")
# interleaved copies: ray i of copy c at position i * rep + c (the copies of a ray are neighbours: same tile for rep <= tile_w)
