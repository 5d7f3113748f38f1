#!/usr/bin/env python3
"""One-off: the two-rays-per-wave UV kernel against the one-ray kernel on 5000 rays of the DTU-like camera (bit-identity)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ngf_amd
from ngf_amd import _lib, synth, uvmapping
for prim in ("sphere", "square"):
    net = uvmapping.NeuTex(primitive_type=prim, sample_num=64); net.load_params(synth.uvmapping_params(5, prim))
    cam, dirs = synth.dtu_rays(600, 800)
    pick = (synth.hash_uniform(2, 1, (5001,)) * np.float32(dirs.shape[0])).astype(np.int64)
    cp, rd = torch.from_numpy(cam)[None].cuda(), torch.from_numpy(dirs[pick])[None].cuda()
    U = torch.from_numpy(synth.hash_uniform(2, 2, (1, 5001, 64))).cuda()
    bg = torch.tensor([[0.2, 0.5, 0.8]]).cuda()
    two = net(cp, rd, bg, jitter_u=U)
    os.environ["NGF_UV_TILES"] = "1"; _lib.knobs_from_env()
    one = net(cp, rd, bg, jitter_u=U)
    os.environ.pop("NGF_UV_TILES"); _lib.knobs_from_env()
    print(prim, "bit-identical:", bool(torch.equal(two["color"], one["color"]) and torch.equal(two["transmittance"], one["transmittance"])),
          "mean colour", float(two["color"].mean()), "finite", bool(torch.isfinite(two["color"]).all()))
