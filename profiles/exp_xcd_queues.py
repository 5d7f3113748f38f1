"""A/B of the per-XCD tile queues (knob xcd = 0 / 1): frame time of the 800x800 presets, TriPlane level 2 / level 1 and InfoInv.
    python profiles/exp_xcd_queues.py > profiles/r03_xcd_queues.txt     (PMC of R2 with / without: profiles/collect_all_r03.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ngf_amd  # noqa
from ngf_amd import _lib, cases, rays as nrays, synth
rays = nrays.generate_rays(800, 800, nrays.blender_focal(800), synth.lookat_pose())

def timed(fn, n=12):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

print("# median kernel ms of one 800x800 frame, S = 192: single tile queue | one queue per XCD with stealing")
for model, preset, flags in (("triplane", "R0", {"bake": True}), ("triplane", "R1", {"bake": True}), ("triplane", "R2", {"bake": True}), ("triplane", "R1", {}),
                             ("triplane", "R2", {}), ("triplane", "R2", {"bake": True, "split_bf16": True}), ("infoinv", "R1", {})):
    g, params, step = cases.big_case(model, preset)
    f = cases.field_for_case(g, params, None, **flags)
    kw = {"iteration": 30001} if model == "triplane" else {"infoinv": True}
    res = []
    for x in (0, 1):
        with _lib.knobs(xcd=x):
            res.append(timed(lambda: f(rays, N_samples=192, white_bg=True, **kw)))
    print(f"{model} {preset} {flags}: {res[0]:.3f} | {res[1]:.3f} ms  ({100 * (res[0] / res[1] - 1):+.1f} %)", flush=True)
    f.release()
