#!/usr/bin/env python3
"""Cycle breakdown of the specialised kernel (ngf_render_pc.hpp, profile build): s_memtime stamps around the march iteration, the
ring-space wait, the sections of the shade pass and the shade waves' idle polling, summed per role and divided by the wave count.
    NGF_WAVES=124|88 python profiles/exp_sections_pc.py [R1|R2 ...]"""
import os, sys
os.environ["NGF_PROFILE"] = "1"; os.environ["NGF_KERNEL"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ngf_amd
from ngf_amd import _lib, cases, rays as nrays, synth
_lib.knobs_from_env()
nm, ns = (8, 8) if os.environ.get("NGF_WAVES") == "88" else (12, 4)
names = ["march iteration", "shade: ring read + setup + gather0 issue", "shade: wait plane 0 + interpolate", "shade: layer-1 MFMAs (144)",
         "shade: layer 2 (64 MFMAs)", "shade: owner collect", "shade: layer 3 (VALU dot + 2 cross-lane adds + sigmoid)"]
rays = nrays.generate_rays(800, 800, nrays.blender_focal(800), synth.lookat_pose())
for preset in (sys.argv[1:] or ["R1", "R2"]):
    g, params, step = cases.big_case("triplane", preset)
    f = cases.field_for_case(g, params, None, device="cuda")
    f(rays, N_samples=192, iteration=30001)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(); f(rays, N_samples=192, iteration=30001, collect_stats=True); ev[1].record(); torch.cuda.synchronize()
    st = f.last_stats.cpu().numpy().astype(np.float64)
    iters = st[0] / 64.0
    print(f"preset {preset}, {nm} march + {ns} shade waves per CU: kernel {ev[0].elapsed_time(ev[1]):.2f} ms (profile build), {st[2]:.0f} passes, ~{iters:.0f} march iterations (evaluated samples / 64)")
    print(f"   march waves: {st[4] / (256 * nm) / 1e6:7.2f} M cycles marching per wave ({st[4] / iters:7.0f} per iteration), {st[11] / (256 * nm) / 1e6:6.2f} M waiting for ring space")
    tot = st[5:11].sum()
    for k in range(1, 7):
        print(f"   {names[k]:58s} {st[4 + k] / st[2]:9.0f} cycles per pass")
    print(f"   shade waves: {tot / (256 * ns) / 1e6:7.2f} M cycles in passes per wave ({tot / st[2]:.0f} per pass), {st[12] / (256 * ns) / 1e6:6.2f} M idle (polling)")
    f.release()
