#!/usr/bin/env python3
"""Cycle breakdown of the fused kernel per wave (NGF_PROFILE build of the default policy): s_memtime stamps around the
march step and the sections of the shade pass, summed over every wave's life and divided by the wave count."""
import os, sys
os.environ["NGF_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
from ngf_amd import _lib, synth
_lib.knobs_from_env()          # NGF_PROFILE above -> ngf_debug_set("profile", 1)
names = ["march steps", "shade: ring read + setup + gather0/view MFMA issue", "shade: wait plane 0 + interpolate",
         "shade: layer-1 MFMAs (160)", "shade: layer 2 (64 MFMAs)", "shade: result list + collect",
         "shade: layer 3 (VALU dot + 2 cross-lane adds + sigmoid)"]
for preset in ("R1",):
    for bd in (False,):
        g, params, step = big_case("triplane", preset)
        f = field_for_case(g, params, None, device="cuda", bake=bd)
        rays = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
        f(rays, N_samples=192, iteration=30001)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record(); f(rays, N_samples=192, iteration=30001, collect_stats=True); ev[1].record(); torch.cuda.synchronize()
        st = f.last_stats.cpu().numpy().astype(np.float64)
        waves = 256 * 12
        print(f"preset {preset} bake_density={bd}: kernel {ev[0].elapsed_time(ev[1]):.2f} ms (profiled build), passes {st[2]:.0f}, steps/wave {640000/64*192/waves:.0f}")
        tot = st[4:11].sum()
        for k in range(7):
            per = st[4 + k] / (st[2] if k else (640000 / 64 * 192))
            print(f"   {names[k]:58s} {100*st[4+k]/tot:5.1f} %  {per:9.0f} cycles per {'pass' if k else 'step'}")
        print(f"   sum = {tot/waves/1e6:.2f} M cycles per wave")
