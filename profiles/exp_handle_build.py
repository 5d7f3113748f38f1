#!/usr/bin/env python3
"""Experiment: cost of (re)building the eval handle (ngf_field_create: texture packing + MLP image) after a parameter change."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
for model in ("triplane", "infoinv"):
    g, params, step = big_case(model, "R1")
    for bake in ((0, 3) if model == "triplane" else (0,)):
        f = field_for_case(g, params, None, bake=bool(bake & 1), bake_color=bool(bake & 2))
        f.handle(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            f._handle_key = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            f.handle(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"{model} bake={bake}: handle rebuild {np.median(ts) * 1e3:.2f} ms")
# ... and with a 256^3 alpha mask on the field (round 6: the create path also builds the mask's cell bytes -- 17 MB -- and its block image)
g, params, step = big_case("triplane", "R1")
f = field_for_case(g, params, None, bake=True, bake_color=True)
f.updateAlphaMask((256, 256, 256))
f.handle(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    f._handle_key = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f.handle(); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"triplane bake=3 + 256^3 alpha mask: handle rebuild {np.median(ts) * 1e3:.2f} ms")
