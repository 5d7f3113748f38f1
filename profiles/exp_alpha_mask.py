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
# filtering_rays (FieldBase.py:218-246; TriPlane/main.py:332-335 calls it on ALL training rays after an alpha-mask update) on ten 800 x 800 views
# (6.4 M rays, N_samples = 256) through the mask just built and through an object-like one (ball of radius 0.8)
from ngf_amd import rays as nrays, synth, triplane
views = [nrays.generate_rays(800, 800, nrays.blender_focal(800), synth.lookat_pose(azim_deg=36.0 * k)) for k in range(10)]
allrays = torch.cat(views, 0)
allrgbs = torch.zeros((allrays.shape[0], 3), device=allrays.device)
for tag in ("own mask", "ball mask"):
    if tag == "ball mask":
        ax = torch.linspace(-1.5, 1.5, 128)
        zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
        f.alphaMask = triplane.AlphaGridMask("cuda", torch.tensor(np.asarray(g["aabb"], np.float32)), ((xx ** 2 + yy ** 2 + zz ** 2) < 0.64).float().cuda())
        f.invalidate()
    f.handle()
    flags = torch.empty((allrays.shape[0],), device="cuda", dtype=torch.uint8)
    import ctypes as C
    from ngf_amd import _lib
    def go():
        _lib.check(_lib.lib().ngf_field_ray_filter(f.handle(), allrays.data_ptr(), allrays.shape[0], 256, flags.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    go(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); go(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"ngf_field_ray_filter, {allrays.shape[0]} rays x 256 samples, {tag}: {np.median(ts) * 1e3:.2f} ms = {allrays.shape[0] / np.median(ts) / 1e6:.0f} Mray/s, kept {float(flags.float().mean()):.3f}")
