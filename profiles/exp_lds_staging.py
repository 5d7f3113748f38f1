#!/usr/bin/env python3
"""Experiment (BASELINE north_star / SURVEY section 7 "measure both"): the LDS-staged texture strips of csrc/ngf_stage.hpp next to
the gather form on the 800x800 frame, S = 192: gauge on (headline path: gauge strips staged) and gauge off (density strips staged,
8 waves per CU).  Prints ms per frame, the share of march iterations whose rectangles fitted the strips, and bit-identity."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ngf_amd
from ngf_amd import _lib, cases, rays as nrays, synth
rays = nrays.generate_rays(800, 800, nrays.blender_focal(800), synth.lookat_pose())


def timed(fn, n=8):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for preset in (sys.argv[1:] or ["R0", "R1"]):
    g, params, step = cases.big_case("triplane", preset)
    f = cases.field_for_case(g, params, None, device="cuda")
    for it, label in ((30001, "gauge on "), (-1, "gauge off")):
        ref = None
        for waves in (12, 8):
            for stage in (0, 1):
                with _lib.knobs(stage=stage, waves=waves, kernel=0):
                    ms = timed(lambda: f(rays, N_samples=192, iteration=it))
                    out = f(rays, N_samples=192, iteration=it, collect_stats=True)
                    st = f.last_stats.cpu().numpy().astype(np.float64)
                if ref is None:
                    ref = out
                same = torch.equal(ref["rgb_map"], out["rgb_map"]) and torch.equal(ref["depth_map"], out["depth_map"])
                iters = st[0] / 64.0
                print(f"{preset} {label} waves/CU {waves:2d} {'LDS strips' if stage else 'gathers   '}: {ms:7.3f} ms = {640000 / ms / 1e3:6.1f} Mray/s"
                      + (f", {100 * st[13] / max(iters, 1):5.1f} % of ~{iters:.0f} march iterations from LDS strips" if stage else "") + f", bit-identical {same}", flush=True)
    f.release()
