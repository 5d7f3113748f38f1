#!/usr/bin/env python3
"""Experiment: updateAlphaMask((256,256,256)) (FieldBase.py:180-216) on the headline field: wall time per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
g, params, step = big_case("triplane", "R1")
f = field_for_case(g, params, None)
for _ in range(2):
    f.alphaMask = None
    aabb = f.updateAlphaMask((256, 256, 256))
torch.cuda.synchronize()
ts = []
for _ in range(5):
    f.alphaMask = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    aabb = f.updateAlphaMask((256, 256, 256))
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"updateAlphaMask(256^3): {np.median(ts) * 1e3:.2f} ms per call, new aabb {aabb.cpu().numpy().round(3).tolist()}, occupied {float(f.alphaMask.alpha_volume.mean()):.3f}")
