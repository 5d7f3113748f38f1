#!/usr/bin/env python3
"""Experiment: throughput of the shade pass alone (ngf_field_decode_rgb) on random samples -- how much of the fused
kernel's per-pass time is the colour MLP itself?  Run on the GPU box: python profiles/exp_decode_throughput.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case

n = 8_000_000
torch.manual_seed(0)
dev = "cuda"
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
for coherent in (0, 1):
    if coherent:   # spatially coherent samples: sorted along a space-filling-ish order (cache-friendly, like rays)
        base = torch.rand(n // 64, 1, 6, device=dev) * 1.8 - 0.9
        coords = (base + 0.01 * torch.randn(n // 64, 64, 6, device=dev)).reshape(-1, 6).contiguous()
    else:
        coords = torch.rand(n, 6, device=dev) * 1.8 - 0.9
    for bc in (False, True):
        g, params, step = big_case("triplane", "R1")
        f = field_for_case(g, params, None, device=dev, bake=False, bake_color=bc)
        f.decode_rgb(coords[:1024], dirs[:1024])
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(3):
            f.decode_rgb(coords, dirs)
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 3
        print(f"coherent={coherent} bake_color={bc}: {n / ms / 1e3:.1f} Msample/s  ({ms:.2f} ms per {n/1e6:.0f} M samples; "
              f"29.6 M samples (one R1 frame) would take {29.6e6 / n * ms:.2f} ms)")
        f.release()
