"""Seeded workloads of the bench and the tests: geometry + parameters of SURVEY.md section 8 D2 and the field built on them.
Product-side only (no oracle, no tests/): bench.py builds its GPU fields from here."""
from __future__ import annotations

import numpy as np

from . import geometry, synth


def big_case(model="triplane", preset="R1", res=256, seed=3):
    """The headline geometry: +-1.5 box, 256^3 grid, 256^2 planes."""
    g = {"model": np.array(model), "aabb": np.array([[-1.5] * 3, [1.5] * 3], np.float32), "grid": np.array([256] * 3),
         "near_far": np.array([2.0, 6.0], np.float32), "step_ratio": np.float32(0.5), "distance_scale": np.float32(25),
         "thr": np.float32(1e-4)}
    hw = ((res, res),) * 3
    if model == "triplane":
        params = synth.triplane_params(seed, hw, (256, 256), preset=preset)
    else:
        params = synth.infoinv_params(seed, hw, preset=preset)
    step = geometry.step_size(g["aabb"], g["grid"], 0.5)
    return g, params, step


def field_for_case(g, params, mask, device="cuda", bake=False, bake_color=False, no_fold=False, split_bf16=False):
    """Build the ngf_amd field (HIP path) for a case dict (a golden fixture or big_case) and a parameter dict."""
    import torch
    from . import infoinv, triplane
    aabb = torch.tensor(np.asarray(g["aabb"], np.float32))
    kw = dict(near_far=[float(v) for v in g["near_far"]], alphaMask_thres=1e-4, distance_scale=float(g["distance_scale"]),
              rayMarch_weight_thres=float(g["thr"]), step_ratio=float(g["step_ratio"]))
    grid = [int(v) for v in g["grid"]]
    if str(g["model"]) == "triplane":
        f = triplane.TriPlane(aabb, grid, device, gauge_start=0, bake_density=bake, bake_color=bake_color, no_fold=no_fold, split_bf16=split_bf16, **kw)
    else:
        f = infoinv.TriPlane(aabb, grid, device, split_bf16=split_bf16, **kw)
    f.load_params(params)
    if mask is not None:
        bits, dhw, maabb = mask
        n = int(np.prod(dhw))
        vol = torch.from_numpy(np.unpackbits(bits)[:n].reshape(dhw).astype(np.float32))
        f.alphaMask = triplane.AlphaGridMask(device, torch.tensor(np.asarray(maabb, np.float32)), vol.to(device))
    return f
