#!/usr/bin/env python3
"""One-off: the WHOLE 800x800 / S=192 frame (640 000 rays) through the C oracle (oracle/ngf_oracle.c, OpenMP on the box's host
cores) against the HIP path, for the headline preset and two others: max abs / max rel (atol 1e-6) / PSNR over every pixel,
and the pixels beyond rtol 1e-4 + atol 1e-5."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case, oracle_for_case
from ngf_amd import synth

rays = synth.lookat_rays(800, 800)
# round 5: the module's default level (3) and, with SPLIT=1, level 3 with layer 2 on the bf16 pipe (TriPlane rows); LEVEL=1 is round 1's run
LEVEL, SPLIT = int(os.environ.get("LEVEL", "3")), bool(int(os.environ.get("SPLIT", "0")))
print(f"TriPlane level {LEVEL}{' + layer 2 as split bf16 products' if SPLIT else ''}; every pixel of the 800x800 / S=192 frame against the C oracle", flush=True)
for model, preset in (("triplane", "R1"), ("triplane", "R2"), ("infoinv", "R1")):
    if SPLIT and model != "triplane":
        continue
    g, params, step = big_case(model, preset)
    g["gauge_on"] = np.array(1); g["infoinv"] = np.array(1)
    orc = oracle_for_case(g, params, step, None)
    f = field_for_case(g, params, None, bake=LEVEL >= 2, bake_color=LEVEL >= 3, split_bf16=SPLIT) if model == "triplane" else field_for_case(g, params, None)
    kw = {"iteration": 30001} if model == "triplane" else {"infoinv": True}
    out = f(torch.from_numpy(rays).cuda(), N_samples=192, white_bg=True, **kw)
    rgb, depth = out["rgb_map"].cpu().numpy(), out["depth_map"].cpu().numpy()
    t0 = time.perf_counter()
    o_rgb, o_depth = orc.render(rays, 192, white_bg=True, threads=min(128, os.cpu_count() or 1))
    dt = time.perf_counter() - t0
    for name, a, b in (("rgb", rgb, o_rgb), ("depth", depth, o_depth)):
        d = np.abs(a.astype(np.float64) - b)
        mse = float((d ** 2).mean())
        bad = d > (1e-5 + 1e-4 * np.abs(b))
        print(f"{model} {preset} {name:5s}: max abs {d.max():.3e}  max rel {(d / (np.abs(b) + 1e-6)).max():.3e}  PSNR {(-10 * np.log10(mse)) if mse else 999:.1f} dB  "
              f"beyond rtol 1e-4 + atol 1e-5: {int(bad.sum())} of {bad.size}   (oracle {dt:.1f} s on the host)", flush=True)
