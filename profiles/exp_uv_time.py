#!/usr/bin/env python3
"""UV-Mapping kernel timing on 76 800 rays x 64 samples (sphere gauge): ms, Mray/s, executed TFLOP/s."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np, ngf_amd
from ngf_amd import synth, uvmapping
net = uvmapping.NeuTex(primitive_type="sphere", sample_num=64, device="cuda"); net.load_params(synth.uvmapping_params(5, "sphere"))
cam, dirs = synth.dtu_rays(600, 800, rows=(252, 348))
cam_t, dirs_t = torch.from_numpy(cam)[None], torch.from_numpy(dirs)[None].cuda()
U = torch.rand((1, dirs.shape[0], 64), device="cuda")
net(cam_t, dirs_t, None, jitter_u=U); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(3): net(cam_t, dirs_t, None, jitter_u=U)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 3
net(cam_t, dirs_t, None, jitter_u=U, collect_stats=True); us = net.last_stats.cpu().numpy().astype(float)
fl = us[1] * 16 * 2 * 1334592.0
print("UV %.1f ms  %.3f Mray/s  executed %.1f TFLOP/s (%.1f%% of 157.3)" % (ms, dirs.shape[0] / ms / 1e3, fl / (ms * 1e-3) / 1e12, 100 * fl / (ms * 1e-3) / 157.3e12))
