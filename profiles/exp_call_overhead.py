#!/usr/bin/env python3
"""Experiment: host-side cost of one field(rays) call (Python mirror + ctypes + launch) with a trivial 8-ray workload, and
the reference-style renderer loop over 4096-ray chunks of a frame against the single launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
from ngf_amd import synth
g, params, step = big_case("triplane", "R1")
f = field_for_case(g, params, None)
rays = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
small = rays[:8].contiguous()
for _ in range(10): f(small, N_samples=192, iteration=30001)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): f(small, N_samples=192, iteration=30001)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"8-ray call: {(t1 - t0) * 1e3:.1f} us issue per call, {(t2 - t0) * 1e3:.1f} us per call incl. drain")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [f(rays[i:i + 4096], N_samples=192, iteration=30001) for i in range(0, rays.shape[0], 4096)]
    torch.cuda.synchronize(); t_chunks = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f(rays, N_samples=192, iteration=30001)
    torch.cuda.synchronize(); t_one = time.perf_counter() - t0
print(f"frame as 157 chunks of 4096 rays: {t_chunks * 1e3:.2f} ms; as one launch: {t_one * 1e3:.2f} ms")
