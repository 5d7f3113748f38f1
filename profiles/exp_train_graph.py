#!/usr/bin/env python3
"""Would a HIP graph of the WHOLE training iteration (backward + Adam + the jitter draw, no Python in between) shorten it?  One Trainer.step
captured with torch.cuda.graph and replayed, against the same step launched normally -- both from the SAME parameter / moment state and with the
SAME frozen step counters (a captured step has its learning rates and Adam bias corrections baked in; left alone the two runs would train
different fields and their active-sample counts, hence their times, drift apart).  A timing experiment, not a training loop."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ngf_amd  # noqa: F401
from ngf_amd import cases, synth, train

dev = "cuda"
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device=dev)
S = int(f.nSamples)
frame = synth.lookat_rays(800, 800)
pick = (synth.hash_uniform(9, 1, (4096,)) * np.float32(frame.shape[0])).astype(np.int64)
rays = torch.from_numpy(frame[pick]).to(dev)
tgt = torch.from_numpy(synth.hash_uniform(9, 2, (4096, 3))).to(dev)
tr = train.Trainer(f, batch_size=4096, max_samples=S)
for it in range(3):
    tr.step(rays, tgt, it, N_samples=S)
torch.cuda.synchronize()
state = ([p.detach().clone() for p in tr.params], [m.clone() for m in tr.exp_avg], [v.clone() for v in tr.exp_avg_sq], list(tr.steps), list(tr.lr))


def restore():
    with torch.no_grad():
        for p, q in zip(tr.params, state[0]): p.copy_(q)
        for p, q in zip(tr.exp_avg, state[1]): p.copy_(q)
        for p, q in zip(tr.exp_avg_sq, state[2]): p.copy_(q)
    tr.steps, tr.lr = list(state[3]), list(state[4])
    tr.params_changed()
    tr.backward(rays, tgt, S, iteration=3)          # re-packs the planes (a plain call); no optimizer step
    torch.cuda.synchronize()


def frozen_step():
    tr.steps, tr.lr = list(state[3]), list(state[4])          # every step is "iteration 3" as in the captured graph
    tr.step(rays, tgt, 3, N_samples=S)


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    restore()
    ms = timed(frozen_step)
    print(f"launched from Python : {ms:.3f} ms / iteration, {tr.last_active} active samples after 20 iterations")
    restore()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        frozen_step()
    torch.cuda.synchronize()
    restore()
    graph = torch.cuda.CUDAGraph()
    tr.steps, tr.lr = list(state[3]), list(state[4])
    with torch.cuda.graph(graph, stream=side):
        tr.step(rays, tgt, 3, N_samples=S)
    torch.cuda.synchronize()
    restore()
    ms = timed(graph.replay)
    print(f"one graph per step   : {ms:.3f} ms / iteration, {tr.last_active} active samples after 20 iterations")
    del graph
