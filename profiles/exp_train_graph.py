#!/usr/bin/env python3
"""Would a HIP graph shorten the training iteration?  One Trainer.step captured with torch.cuda.graph (the library's launches, its two aux streams
and their event forks / joins are all capturable: no allocation, no host sync inside a step) and replayed, next to the same step launched normally.
The captured step has its iteration number (learning rate, Adam bias corrections) frozen: a timing experiment, not a training loop."""
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


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


it = [3]


def eager():
    tr.step(rays, tgt, it[0], N_samples=S)
    it[0] += 1


print(f"eager launches : {timed(eager, 20):.3f} ms / iteration ({tr.last_active} active samples)")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(side):
        tr.step(rays, tgt, it[0], N_samples=S)          # warm-up on the capture stream
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        tr.step(rays, tgt, it[0], N_samples=S)
    torch.cuda.synchronize()
    print(f"graph replay   : {timed(graph.replay, 20):.3f} ms / iteration")
    print(f"eager again    : {timed(eager, 20):.3f} ms / iteration")
    print(f"graph replay   : {timed(graph.replay, 20):.3f} ms / iteration")
except Exception as e:      # noqa: BLE001
    print("capture failed:", repr(e)[:400])
