#!/usr/bin/env python3
"""Experiment: where a pass of train_color_bwd_kernel spends its clocks (ngf_debug_set("ablate", 1 << 20) makes every wave add
its per-section cycle counts to eight counters; ngf_train_debug_sections reads them).  Same batch as profiles/workload.py train_R1."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ngf_amd  # noqa: F401
from ngf_amd import _lib, cases, synth, train

if os.environ.get("NGF_LIB"):          # an experiment build (make -C neural-gauge-fields_amd/csrc exp NAME=... DEFS=...)
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
L = _lib.lib()
L.ngf_train_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device="cuda")
S = int(f.nSamples)
frame = synth.lookat_rays(800, 800)
pick = (synth.hash_uniform(9, 1, (4096,)) * np.float32(frame.shape[0])).astype(np.int64)
rays = torch.from_numpy(frame[pick]).cuda()
tgt = torch.from_numpy(synth.hash_uniform(9, 2, (4096, 3))).cuda()
tr = train.Trainer(f, batch_size=4096, max_samples=S)
for it in range(3):
    tr.step(rays, tgt, it, N_samples=S)
names = ["rows -> tiles (H1, H2), d3", "d2 + layer-2 backward (MFMA)", "D3/D2/D1 rows out, bias sums", "df = W1'^T d1 (MFMA)", "coords, taps, d loss / d t",
         "DF rows out, places in the bins"]
with _lib.knobs(ablate=(1 << 20) | (1 << 21) | int(os.environ.get('EXTRA_ABLATE', '0'))):
    out = (C.c_uint64 * 16)()
    _lib.check(L.ngf_train_debug_sections(tr._h, out))
    n = 5
    for it in range(n):
        tr.step(rays, tgt, 3 + it, N_samples=S)
    _lib.check(L.ngf_train_debug_sections(tr._h, out))
passes = out[7]
tot = sum(out[k] for k in range(6))
print(f"{passes / n:.0f} passes of 16 samples per iteration; clocks per pass (100 MHz counter ticks x 1 -- relative shares matter):")
for k in range(6):
    print(f"  {names[k]:40s} {out[k] / passes:10.1f}   {100.0 * out[k] / tot:5.1f} %")
print(f"  {'total':40s} {tot / passes:10.1f}")
if out[15]:
    print(f"train_bin_scatter_kernel: {out[15] / n:.0f} units per iteration; clocks per unit: zero tile + records {out[11] / out[15]:.1f}, "
          f"gradient rows + LDS sums {out[12] / out[15]:.1f}, flush (atomics) {out[13] / out[15]:.1f}")
print(f"atomic line transactions per iteration: density/gauge backward {out[8] / n:.0f}, colour backward {out[9] / n:.0f} "
      f"-> {(out[8] + out[9]) / n / 21.0e9 * 1e3:.3f} ms at the measured 21 G transactions/s (profiles/r02_micro_atomic_cost.txt)")
