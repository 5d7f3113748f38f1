import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ngf_amd
from ngf_amd import _lib, cases, synth
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None)
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
def t(rays, n=4):
    ev=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    f(rays, N_samples=192, white_bg=True, iteration=30001)
    for a,b in ev:
        a.record(); f(rays, N_samples=192, white_bg=True, iteration=30001); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a,b in ev]))
for rep in (1, 2, 4):
    rays = frame.repeat(rep, 1)
    ms = t(rays)
    print(f"{rep} x frame in one launch: {ms:.3f} ms = {rays.shape[0]/ms/1e3:.2f} Mray/s")
for tw in (4, 8, 16):
    with _lib.knobs(tile_w=tw):
        ms = t(frame)
    print(f"tile_w={tw}: {ms:.3f} ms = {frame.shape[0]/ms/1e3:.2f} Mray/s")
