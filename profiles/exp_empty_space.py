#!/usr/bin/env python3
"""Experiment: a scene with an alpha mask (what every trained checkpoint of the reference carries): occupancy = a ball of
radius 0.8 in the +-1.5 box (15 % of the volume), S = 192 and the model's own auto S (884).  Times the frame with and
without the empty-iteration skip / early termination (NGF_ABLATE bits 64 / 32 switch them off; all three are bit-identical)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
from ngf_amd import _lib, synth, triplane

rays = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
g, params, step = big_case("triplane", "R1")
f = field_for_case(g, params, None)
D = 128
ax = torch.linspace(-1.5, 1.5, D)
zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
vol = ((xx ** 2 + yy ** 2 + zz ** 2) < 0.8 ** 2).float()
f.alphaMask = triplane.AlphaGridMask("cuda", torch.tensor(np.asarray(g["aabb"], np.float32)), vol.cuda())
f._handle_key = None
LABEL = {"96": "no skip, no early termination", "32": "empty-iteration skip only      ", "": "skip + early termination       "}
for S in (192, -1):
    for mode in ("96", "32", ""):
        if mode: os.environ["NGF_ABLATE"] = mode
        else: os.environ.pop("NGF_ABLATE", None)
        _lib.knobs_from_env()          # the library itself never reads the environment (ngf_debug_set)
        for _ in range(2): out = f(rays, N_samples=S, iteration=30001)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record(); out = f(rays, N_samples=S, iteration=30001); b.record()
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        f(rays, N_samples=S, iteration=30001, collect_stats=True); torch.cuda.synchronize()
        st = f.last_stats.cpu().numpy()
        if mode == "96": ref = (out["rgb_map"].clone(), out["depth_map"].clone())
        else: assert torch.equal(ref[0], out["rgb_map"]) and torch.equal(ref[1], out["depth_map"]), "not bit-identical"
        print(f"S={S if S > 0 else f.nSamples}: {LABEL[mode]}: {ms:7.3f} ms  ({640000 / ms / 1e3:6.1f} Mray/s), "
              f"{st[0] / 640000:6.1f} samples/ray evaluated, {st[1] / 640000:5.1f} active")
