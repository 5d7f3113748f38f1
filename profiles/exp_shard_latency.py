#!/usr/bin/env python3
"""Experiment: kernel time of ONE 8-GPU shard (100 rows = 80 000 rays of the 800x800 R1 frame) vs the full frame,
for the waves-per-CU / steps-in-flight knobs.  Strong scaling at 8 GPUs is bounded by this latency."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
from helpers import big_case, field_for_case
import ngf_amd
from ngf_amd import _lib, synth
_lib.knobs_from_env()
g, params, step = big_case("triplane", "R1")
bd = bool(int(os.environ.get("BD", "0")))
f = field_for_case(g, params, None, device="cuda", bake=bd)
for rows in ((350, 450), (0, 100), (0, 800)):
    rays = torch.from_numpy(synth.lookat_rays(800, 800, rows=rows)).cuda()
    for _ in range(2): f(rays, N_samples=192, iteration=30001)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); f(rays, N_samples=192, iteration=30001); b.record()
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in ev])
    print(f"  rows {rows}: {ms:.3f} ms  ({rays.shape[0] / ms / 1e3:.1f} Mray/s)")
''' % (ROOT, ROOT)
if os.environ.get("ONLY_DEFAULT"):          # the product library's only configuration: level 2, 12 waves, one step per lane
    env = dict(os.environ, BD="1")
    print("bake_density=1 (level 2, the module default), product library")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout, r.stderr[-300:] if r.returncode else "")
    sys.exit(0)
for bd in ("0", "1"):
    for w, ns in (("12", "1"), ("8", "1"), ("8", "2")):
        env = dict(os.environ, NGF_WAVES=w, NGF_NSTEP=ns, BD=bd)      # these knobs select the experiment library (libngf_hip_exp.so)
        print(f"bake_density={bd} waves={w} nstep={ns}")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(r.stdout, r.stderr[-300:] if r.returncode else "")
