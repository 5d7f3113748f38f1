#!/usr/bin/env python3
"""Experiment: one TriPlane training iteration at the reference's training shape -- 4096 random rays of
the 800x800 frame (args.batch_size), 256^2 planes, gauge on, nSamples = the model's auto value (884 on a 256^3 grid at
step_ratio 0.5) -- on the MI355X (ngf_amd.train.Trainer) next to the reference's own way on this box's CPU cores (autograd
of the eager port, oracle/train.py).  Per-kernel times come from rocprofv3 (profiles/r01_train_kernel_stats.txt)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
import ngf_amd
from ngf_amd import synth, train

steps = int(os.environ.get("STEPS", "20"))
cpu = int(os.environ.get("CPU", "1"))
for preset in ("R1", "R2"):
    g, params, step = big_case("triplane", preset)
    f = field_for_case(g, params, None)
    S = int(os.environ.get("S", f.nSamples))
    frame = synth.lookat_rays(800, 800)
    pick = (synth.hash_uniform(9, 1, (4096,)) * np.float32(frame.shape[0])).astype(np.int64)
    rays = torch.from_numpy(frame[pick]).cuda()
    tgt = torch.from_numpy(synth.hash_uniform(9, 2, (4096, 3))).cuda()
    tr = train.Trainer(f, batch_size=4096, max_samples=S)
    for it in range(3):
        loss = tr.step(rays, tgt, it, N_samples=S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        loss = tr.step(rays, tgt, 3 + it, N_samples=S)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"preset {preset}: S={S} active samples/step {tr.last_active} ({tr.last_active / 4096:.1f} per ray), "
          f"{ms:.2f} ms/iteration = {1e3 / ms:.1f} it/s, {4096 / ms / 1e3:.3f} Mray/s trained, loss {loss.item():.5f}, scratch {tr.scratch_bytes() / 2**30:.2f} GiB")
    if cpu and preset == "R1":
        from oracle import train as otrain
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        orc = otrain.EagerTrainer(params, g["aabb"], step, g["near_far"], float(g["distance_scale"]), float(g["thr"]))
        jit = torch.rand(4096)
        t0 = time.perf_counter()
        orc.gradients(rays.cpu(), tgt.cpu(), S, jit, True, 5)
        print(f"   CPU autograd of the eager port, {torch.get_num_threads()} threads: {time.perf_counter() - t0:.2f} s per forward+backward (no optimiser)")
