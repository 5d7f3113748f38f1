#!/usr/bin/env python3
"""Where a wave of uv_render_kernel spends its cycles (BASELINE config 4: 76 800 DTU-camera rays x 64 samples).

    make -C neural-gauge-fields_amd/csrc expuv NAME=uvsec DEFS=-DNGF_EXP_UV_SECTIONS=1
    NGF_LIB=neural-gauge-fields_amd/csrc/build/exp/uvsec/libngf_hip.so python profiles/exp_uv_sections.py

The experiment build reads s_memtime at the section boundaries of ngf_uv.hpp (never inside the k loop) and adds every wave's sums to
UvArgs::stats[2..9]: 0 = layer prologue (bias + the first four k-steps of weights ARRIVED), 1 = the k loops, 2 = activation + LDS store of a layer's
outputs, 3 = positional-encoding inputs, 4 = the <= 3-unit output layers, 5 = the three networks of a pass (0..4 and the glue between them),
6 = MFMAs issued by the k loops, 7 = the wave's life.  Product-side imports only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ngf_amd  # noqa: F401
from ngf_amd import _lib, rays as nrays, synth, uvmapping

if os.environ.get("NGF_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
dev = "cuda"
split = bool(int(os.environ.get("SPLIT", "0")))
net = uvmapping.NeuTex(primitive_type="sphere", sample_num=64, device=dev, split_bf16=split)
net.load_params(synth.uvmapping_params(5, "sphere"))
v = synth.DTU_VIEW0
dirs = nrays.generate_rays_dtu(600, 800, v["focal"], v["princpt"], v["rot"], rows=(252, 348))[None]
cam = torch.tensor(v["campos"], dtype=torch.float32)[None]
U = torch.rand((1, dirs.shape[1], 64), device=dev)
run = lambda **kw: net(cam, dirs, None, jitter_u=U, **kw)
run()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
for a, b in ev:
    a.record(); run(); b.record()
torch.cuda.synchronize()
ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
run(collect_stats=True)
st = net.last_stats.cpu().numpy().astype(np.float64)
fl = st[1] * 16 * 2 * 1334592.0
print(f"lib {os.path.basename(os.path.dirname(_lib.SO_PATH))}: {ms:.2f} ms = {dirs.shape[1] / ms / 1e3:.3f} Mray/s; in-cube {st[0] / dirs.shape[1]:.1f} samples/ray, "
      f"{st[1]:.0f} 16-sample tiles, executed {fl / (ms * 1e-3) / 1e12:.1f} TFLOP/s = {fl / (ms * 1e-3) / 157.3e12:.3f} of the fp32 MFMA peak")
if st[9] > 0:
    life = st[9]
    names = ["layer prologue (bias + 4 k-steps of weights arrived)", "k loops", "activation + LDS store", "positional-encoding inputs", "output layers (<= 3 units)"]
    for i, n in enumerate(names):
        print(f"  section {i} {n:55s} {100 * st[2 + i] / life:6.2f} % of the waves' life")
    print(f"  networks of a pass, all included                                  {100 * st[7] / life:6.2f} %   (glue between the sections: {100 * (st[7] - st[2:7].sum()) / life:.2f} %)")
    print(f"  outside the networks (ray set-up, prefix sums, compaction, compositing) {100 * (life - st[7]) / life:6.2f} %")
    print(f"  k loops: {st[8]:.4g} MFMAs in {st[3]:.4g} cycles = {st[3] / st[8]:.2f} cycles per MFMA (s_memtime ticks; 32 = the matrix pipe's rate)")
