#!/usr/bin/env python3
"""Round 6: where the reference's own loop on the drop-in field spends an iteration (TriPlane/main.py:272-299 at the
reference's batch shape: 4096 rays x 884 samples, 256^2 planes) -- with torch.optim.Adam as the reference writes it and with ngf_amd.optim.Adam.
Per section: host time to ENQUEUE it (no synchronisation inside an iteration except the forward's own active-count read) and, in a second pass
with a device synchronisation after every section, its device + host time.

    python profiles/exp_autograd_loop.py [torch|ngf|both] [iterations]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ngf_amd  # noqa: F401
from ngf_amd import cases, optim, synth, train

which = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda"
frame = synth.lookat_rays(800, 800)
pick = (synth.hash_uniform(9, 1, (4096,)) * np.float32(frame.shape[0])).astype(np.int64)
rays = torch.from_numpy(frame[pick]).to(dev)
tgt = torch.from_numpy(synth.hash_uniform(9, 2, (4096, 3))).to(dev)

SECTIONS = ("forward", "loss", "zero_grad", "backward", "step", "item")


def run(cls, sync_sections):
    g, params, step = cases.big_case("triplane", "R1")
    f = cases.field_for_case(g, params, None, device=dev, bake=True)
    S = int(f.nSamples)
    opt = (torch.optim.Adam if cls == "torch" else optim.Adam)(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    acc = {k: 0.0 for k in SECTIONS}

    def it(i, timed):
        marks = [time.perf_counter()]

        def mark():
            if sync_sections:
                torch.cuda.synchronize()
            marks.append(time.perf_counter())
        out = f(rays, is_train=True, white_bg=True, N_samples=S, iteration=i); mark()
        rgb_loss = torch.mean((out["rgb_map"] - tgt) ** 2)
        total = rgb_loss + 8e-5 * f.density_L1(); mark()
        opt.zero_grad(); mark()
        total.backward(); mark()
        opt.step(); mark()
        rgb_loss.detach().item(); mark()
        if timed:
            for k, a, b in zip(SECTIONS, marks[:-1], marks[1:]):
                acc[k] += b - a
    for i in range(3):
        it(i, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        it(3 + i, True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    active = f._grad_engine.last_active
    f._grad_engine.release(); f.release()
    return ms, {k: v / iters * 1e3 for k, v in acc.items()}, active


for cls in (("torch", "ngf") if which == "both" else (which,)):
    for sync in (False, True):
        ms, acc, active = run(cls, sync)
        print(f"{cls:5s} Adam, {'synchronised after every section' if sync else 'as the loop runs (enqueue times)  '}: {ms:6.3f} ms/iteration  " +
              "  ".join(f"{k} {v:6.3f}" for k, v in acc.items()) + f"   ({active} active samples)", flush=True)

# the fused trainer next to it
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device=dev, bake=True)
S = int(f.nSamples)
tr = train.Trainer(f, batch_size=4096, max_samples=S)
for i in range(3):
    tr.step(rays, tgt, i, N_samples=S)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters):
    tr.step(rays, tgt, 3 + i, N_samples=S).item()
torch.cuda.synchronize()
print(f"fused Trainer.step (+ .item() per iteration): {(time.perf_counter() - t0) / iters * 1e3:6.3f} ms/iteration", flush=True)
