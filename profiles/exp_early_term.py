#!/usr/bin/env python3
"""Experiment: exact early termination of the march (T below half an ulp of every accumulator).  Bit-identity against the
full march (NGF_ABLATE=32 switches the early exit off) and the time it saves, per preset, on the 800x800 frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
from ngf_amd import _lib, synth

rays = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
for model, preset, bake in (("triplane", "R1", 0), ("triplane", "R2", 0), ("triplane", "R0", 0), ("triplane", "R1", 3), ("infoinv", "R1", 0)):
    g, params, step = big_case(model, preset)
    f = field_for_case(g, params, None, bake=bool(bake & 1), bake_color=bool(bake & 2))
    kw = {"iteration": 30001} if model == "triplane" else {"infoinv": True}
    res = {}
    for mode in ("32", ""):
        if mode: os.environ["NGF_ABLATE"] = mode
        else: os.environ.pop("NGF_ABLATE", None)
        _lib.knobs_from_env()          # the library itself never reads the environment (ngf_debug_set)
        for _ in range(2): out = f(rays, N_samples=192, collect_stats=True, **kw)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); out = f(rays, N_samples=192, **kw); b.record()
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        f(rays, N_samples=192, collect_stats=True, **kw); torch.cuda.synchronize()
        res[mode] = (out["rgb_map"].clone(), out["depth_map"].clone(), ms, f.last_stats[0].item() / rays.shape[0])
    same = torch.equal(res["32"][0], res[""][0]) and torch.equal(res["32"][1], res[""][1])
    print(f"{model} {preset} bake={bake}: full march {res['32'][2]:.3f} ms ({res['32'][3]:.1f} in-box samples/ray evaluated) -> early exit {res[''][2]:.3f} ms "
          f"({res[''][3]:.1f}); bit-identical: {same}")
