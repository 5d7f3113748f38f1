#!/usr/bin/env python3
"""One named GPU workload, run a few times -- what profiles/collect.sh puts under rocprofv3.  Product-side imports only.

    python profiles/workload.py <name> [reps]
names: triplane_R0 | triplane_R1 | triplane_R2 | triplane_R1_bd (bake density) | triplane_R1_bdc (both bakes) | triplane_R1_nofold | infoinv_R1
       (800x800 frame, S = 192, BASELINE configs 2 / 3) | <any triplane name>_S884mask: the reference's own evaluation shape -- renderer(..., N_samples=-1)
       = 884 steps (TriPlane/main.py:94, FieldBase.py:71-72) through the alpha mask updateAlphaMask((256,)*3) builds from the field (main.py:330) --
       | ..._S884ball: the same steps, occupancy = a ball of radius 0.8 (15 % of the box: an object, like a trained lego) | ..._S884lattice / _S884shell: thin walls every 32 cells inside a ball / a thin spherical shell (cluttered occupancy, a bare surface: round 6's empty-space skipping) | uv_sphere (BASELINE config 4: 76 800 DTU-camera rays x 64 samples)
       | train_R1 (4096-ray training iteration)
Optional knobs through the environment of THIS script (mapped to ngf_debug_set): NGF_KERNEL, NGF_TILE_W, NGF_STAGE, ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ngf_amd  # noqa: F401
from ngf_amd import _lib, cases, rays as nrays, synth

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if os.environ.get("NGF_LIB"):          # an experiment build (make -C neural-gauge-fields_amd/csrc exp NAME=... DEFS=...)
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
_lib.knobs_from_env()
dev = "cuda"


def timed(fn, n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


if name.startswith(("triplane", "infoinv")):
    parts = name.split("_")
    model, preset, bake = parts[0], parts[1], (parts[2] if len(parts) > 2 else "")
    g, params, step = cases.big_case(model, preset)
    nofold, split = bake == "nofold", bake.startswith("split") or bake == "bdcs"          # "split" | "splitd" (with baked density) | "bdcs" (level 3 + bf16 layer 2)
    plain = not (nofold or split)
    f = cases.field_for_case(g, params, None, device=dev, bake=model == "triplane" and ((plain and "d" in bake) or bake in ("splitd", "bdcs")),
                             bake_color=model == "triplane" and ((plain and "c" in bake) or bake == "bdcs"), no_fold=nofold, split_bf16=split)
    rays = nrays.generate_rays(800, 800, nrays.blender_focal(800), synth.lookat_pose())
    kw = {"iteration": 30001} if model == "triplane" else {"infoinv": True}
    shape = parts[3] if len(parts) > 3 else ""
    NS = 192
    if shape.startswith("S884"):
        NS = -1                                                   # the model's own nSamples (884 for the 256^3 grid at step_ratio 0.5)
        if shape == "S884mask":
            f.updateAlphaMask((256, 256, 256), **({} if model == "triplane" else {"infoinv": True}))
        elif shape in ("S884lattice", "S884shell"):
            # cluttered occupancy (the adversarial side of the empty-space skipping): thin walls every 32 cells along all three axes inside a ball of radius 1.1
            # (empty cells everywhere, almost no block with 8 clear cells around it) | a thin spherical shell (an object's surface: empty inside and outside)
            from ngf_amd import triplane
            ax = torch.linspace(-1.5, 1.5, 256)
            zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
            r2 = xx ** 2 + yy ** 2 + zz ** 2
            if shape == "S884lattice":
                ii = torch.arange(256)
                wall = (ii % 32) < 2
                vol = (wall[:, None, None] | wall[None, :, None] | wall[None, None, :]) & (r2 < 1.1 ** 2)
            else:
                vol = (r2 < 0.85 ** 2) & (r2 > 0.80 ** 2)
            f.alphaMask = triplane.AlphaGridMask(dev, torch.tensor(np.asarray(g["aabb"], np.float32)), vol.float().to(dev))
            f.invalidate()
        else:
            from ngf_amd import triplane
            ax = torch.linspace(-1.5, 1.5, 128)
            zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
            f.alphaMask = triplane.AlphaGridMask(dev, torch.tensor(np.asarray(g["aabb"], np.float32)), ((xx ** 2 + yy ** 2 + zz ** 2) < 0.8 ** 2).float().to(dev))
            f.invalidate()
    if int(os.environ.get("NGF_ROW_WIDTH", "800")) > 0 and model == "triplane":
        kw["row_width"] = int(os.environ.get("NGF_ROW_WIDTH", "800"))       # the frame is an image: screen-space tile order (round 6), as bench.py and evalout.evaluation run it; NGF_ROW_WIDTH=0: the list's order
    run = lambda: f(rays, N_samples=NS, white_bg=True, **kw)
    with torch.no_grad():
        run(); run()
        ms = timed(run, reps)
        f(rays, N_samples=NS, white_bg=True, collect_stats=True, **kw)
    st = f.last_stats.cpu().numpy().astype(np.float64)
    occ = "" if f.alphaMask is None else f", mask occupancy {float(f.alphaMask.alpha_volume.mean()):.3f}"
    print(f"{name}: {ms:.3f} ms/frame = {640000 / ms / 1e3:.2f} Mray/s; S={NS if NS > 0 else f.nSamples}{occ}; evaluated {st[0] / 640000:.1f} active {st[1] / 640000:.2f} samples/ray, {st[2]:.0f} passes")
elif name in ("uv_sphere", "uv_sphere_split"):
    from ngf_amd import uvmapping
    net = uvmapping.NeuTex(primitive_type="sphere", sample_num=64, device=dev, split_bf16=name.endswith("split"))
    net.load_params(synth.uvmapping_params(5, "sphere"))
    v = synth.DTU_VIEW0
    dirs = nrays.generate_rays_dtu(600, 800, v["focal"], v["princpt"], v["rot"], rows=(252, 348))[None]
    cam = torch.tensor(v["campos"], dtype=torch.float32)[None]
    U = torch.rand((1, dirs.shape[1], 64), device=dev)
    run = lambda: net(cam, dirs, None, jitter_u=U)
    run()
    ms = timed(run, max(2, reps // 2))
    net(cam, dirs, None, jitter_u=U, collect_stats=True)
    us = net.last_stats.cpu().numpy().astype(np.float64)
    fl = us[1] * 16 * 2 * 1334592.0
    print(f"{name}: {ms:.2f} ms = {dirs.shape[1] / ms / 1e3:.3f} Mray/s; in-cube {us[0] / dirs.shape[1]:.1f} samples/ray, executed {fl / (ms * 1e-3) / 1e12:.1f} TFLOP/s")
elif name.startswith("train"):
    from ngf_amd import train
    preset = name.split("_")[1]
    g, params, step = cases.big_case("triplane", preset)
    f = cases.field_for_case(g, params, None, device=dev)
    S = int(f.nSamples)
    frame = synth.lookat_rays(800, 800)
    pick = (synth.hash_uniform(9, 1, (4096,)) * np.float32(frame.shape[0])).astype(np.int64)
    rays = torch.from_numpy(frame[pick]).to(dev)
    tgt = torch.from_numpy(synth.hash_uniform(9, 2, (4096, 3))).to(dev)
    tr = train.Trainer(f, batch_size=4096, max_samples=S)
    it = [0]

    def run():
        tr.step(rays, tgt, it[0], N_samples=S); it[0] += 1
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name}: S={S}, {tr.last_active} active samples, {ms:.3f} ms/iteration")
else:
    raise SystemExit("unknown workload " + name)
