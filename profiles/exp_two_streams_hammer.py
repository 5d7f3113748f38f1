#!/usr/bin/env python3
"""Overlapping launches of ONE field handle (frames on two alternating render streams, as bench.py runs at N > 1): N frames of an 80 000-ray shard, every
frame compared on the device with the serial frame.  Level 3, level 3 + bf16 layer 2 and InfoInv.   python profiles/exp_two_streams_hammer.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngf_amd  # noqa: F401
from ngf_amd import cases, dist as ndist, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda:0")
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda().view(800, 800, 6)
rays = torch.cat([frame[a:b] for a, b in ndist.interleaved_rows(800, 8, 0, 10)]).reshape(-1, 6).contiguous()
MASK = os.environ.get("MASK") == "1"          # MASK=1: the same through an object-like alpha mask at the reference's S = 884 (MaskSkip instantiations: one-byte mask test + empty-space skipping)
NS = -1 if MASK else 192
for label, model, flags, kw in (("level 3", "triplane", dict(bake=True, bake_color=True), dict(iteration=30001, row_width=800)),
                                ("level 3 + split bf16", "triplane", dict(bake=True, bake_color=True, split_bf16=True), dict(iteration=30001, row_width=800)),
                                ("InfoInv", "infoinv", {}, dict(infoinv=True))):
    g, params, step = cases.big_case(model, "R1")
    f = cases.field_for_case(g, params, None, device="cuda", **flags)
    if MASK:
        import numpy as np
        from ngf_amd import triplane
        ax = torch.linspace(-1.5, 1.5, 128)
        zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
        f.alphaMask = triplane.AlphaGridMask("cuda", torch.tensor(np.asarray(g["aabb"], np.float32)), ((xx ** 2 + yy ** 2 + zz ** 2) < 0.64).float().cuda())
        f.invalidate()
        label += " + ball mask, S = 884"
    ref = f(rays, N_samples=NS, white_bg=True, **kw)
    ref = (ref["rgb_map"].clone(), ref["depth_map"].clone())
    streams = ndist.render_streams(dev)
    ring = [(torch.empty_like(ref[0]), torch.empty_like(ref[1])) for _ in range(4)]
    bad = [torch.zeros((), device=dev, dtype=torch.int64) for _ in range(2)]          # one counter per stream (no read-modify-write across streams)
    for k in range(N):
        st = streams[k % 2]
        with torch.cuda.stream(st):
            o = ring[k % 4]
            f(rays, N_samples=NS, white_bg=True, out=o, **kw)
            bad[k % 2] += ((o[0] != ref[0]).any() | (o[1] != ref[1]).any()).to(torch.int64)
    torch.cuda.synchronize()
    print(f"{label}: {N} frames of {rays.shape[0]} rays on two alternating streams, {int(bad[0].item()) + int(bad[1].item())} differ from the serial frame")
    f.release()
