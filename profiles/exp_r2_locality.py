#!/usr/bin/env python3
"""Round 6: what the L2 re-streaming of the level-3 R2 / S = 884 frames costs, and what ray ORDER does about it.

(a) ablation -- the same frame with planes small enough to live in ONE XCD's 4 MB L2 (res 64: 3 x 66^2 x 256 B = 3.3 MB) and in between:
    what the frame would cost if the colour-plane traffic never left the L2;
(b) the ray list in screen-space blocks (BH rows x BW columns, BW a multiple of the 8-ray tile) instead of image rows, with and without
    one tile queue per XCD (knob xcd): a block's rays form a narrow frustum whose footprint on each plane is a strip, a row band's is a fan.
Timing only (the permuted frame holds the same pixels in another order; bit-identity of every ray is checked).

    python profiles/exp_r2_locality.py [triplane_R2 | triplane_R2_S884mask | infoinv_R1_S884mask | triplane_R1 ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ngf_amd  # noqa: F401
from ngf_amd import _lib, cases, rays as nrays, synth

dev = "cuda"
H = W = 800


def timed(fn, n=7):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def block_order(bh, bw):
    """indices of the H x W image in blocks of bh rows x bw columns (blocks row-major, rays inside a block row-major)"""
    idx = np.arange(H * W, dtype=np.int64).reshape(H // bh, bh, W // bw, bw).transpose(0, 2, 1, 3).reshape(-1)
    return torch.from_numpy(idx).to(dev)


def build(name, res=256):
    parts = name.split("_")
    model, preset = parts[0], parts[1]
    shape = parts[2] if len(parts) > 2 else ""
    g, params, step = cases.big_case(model, preset, res=res)
    tri = model == "triplane"
    f = cases.field_for_case(g, params, None, device=dev, bake=tri, bake_color=tri)
    kw = {"iteration": 30001} if tri else {"infoinv": True}
    NS = 192
    if shape == "S884mask":
        NS = -1
        f.updateAlphaMask((256, 256, 256), **({} if tri else {"infoinv": True}))
    return f, kw, NS


names = sys.argv[1:] or ["triplane_R2", "triplane_R2_S884mask", "infoinv_R1_S884mask", "triplane_R1"]
rays = nrays.generate_rays(H, W, nrays.blender_focal(H), synth.lookat_pose())
with torch.no_grad():
    for name in names:
        print(f"== {name}", flush=True)
        # (a) working-set ablation
        for res in (256, 128, 64, 32):
            f, kw, NS = build(name, res)
            ms = timed(lambda: f(rays, N_samples=NS, white_bg=True, **kw))
            f(rays, N_samples=NS, white_bg=True, collect_stats=True, **kw)
            st = f.last_stats.cpu().numpy().astype(np.float64)
            print(f"  planes {res:3d}^2: {ms:8.3f} ms = {640000 / ms / 1e3:7.2f} Mray/s   evaluated {st[0] / 640000:6.1f} active {st[1] / 640000:6.2f} samples/ray  passes {st[2]:.0f}  "
                  f"-> {ms / max(st[2], 1) * 1e6:7.2f} ns/pass", flush=True)
            f.release()
        # (b) ray order
        f, kw, NS = build(name)
        ref = f(rays, N_samples=NS, white_bg=True, **kw)
        base = {}
        for xcd in (0, 1):
            with _lib.knobs(xcd=xcd):
                base[xcd] = timed(lambda: f(rays, N_samples=NS, white_bg=True, **kw))
        print(f"  image rows         : xcd=0 {base[0]:8.3f} ms   xcd=1 {base[1]:8.3f} ms", flush=True)
        for bh, bw in ((8, 8), (16, 16), (32, 32), (40, 40), (80, 80), (100, 200), (100, 400), (200, 800), (400, 400), (16, 800), (32, 800), (800, 8), (800, 16), (800, 32)):
            if H % bh or W % bw:
                continue
            idx = block_order(bh, bw)
            rp = rays[idx].contiguous()
            t = {}
            for xcd in (0, 1):
                with _lib.knobs(xcd=xcd):
                    t[xcd] = timed(lambda: f(rp, N_samples=NS, white_bg=True, **kw))
            out = f(rp, N_samples=NS, white_bg=True, **kw)
            same = bool(torch.equal(out["rgb_map"], ref["rgb_map"][idx]) and torch.equal(out["depth_map"], ref["depth_map"][idx]))
            print(f"  blocks {bh:3d} x {bw:3d}    : xcd=0 {t[0]:8.3f} ms ({t[0] / base[0] - 1:+.1%})   xcd=1 {t[1]:8.3f} ms ({t[1] / base[0] - 1:+.1%})   same bits per ray: {same}", flush=True)
        f.release()
