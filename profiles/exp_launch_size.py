#!/usr/bin/env python3
"""Launch size vs time of the product library (LEVEL=3, the module default since round 4; LEVEL=2: round 3's): contiguous row blocks of the 800x800 R1 frame, S = 192.
4 096 rays = the reference's chunk / training batch (TriPlane/main.py:94,272), 80 000 rays = one rank's shard of an 8-GPU frame.
TILES="auto 4 8" sweeps the tile-width knob next to the library's own choice.  Output: profiles/r04_shard_latency.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ngf_amd  # noqa
from ngf_amd import _lib, synth
if os.environ.get("NGF_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
from ngf_amd.cases import big_case, field_for_case
import hashlib
model = os.environ.get("MODEL", "triplane")
g, params, step = big_case(model, "R1")
LEVEL = int(os.environ.get("LEVEL", "3"))          # 3 = the module default since round 4; 2 = round 3's
f = field_for_case(g, params, None, device="cuda", bake=True, bake_color=LEVEL >= 3)
kw = dict(iteration=30001) if model == "triplane" else dict(infoinv=True)
print(f"library sha256 {hashlib.sha256(open(_lib.SO_PATH,'rb').read()).hexdigest()[:16]}  model {model} level {LEVEL if model == 'triplane' else '-'}  S=192  {torch.cuda.get_device_name(0)}")
full = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
ref = f(full, N_samples=192, **kw)
if os.environ.get("TAIL"):          # narrow tiles per resident wave and width, x 16 (launch_render's tile plan); unset = the library's default
    _lib.check(_lib.lib().ngf_debug_set(b"tail", int(os.environ["TAIL"])))
    print("knob tail =", os.environ["TAIL"])
SIZES = [int(v) for v in os.environ.get("SIZES", "4096 40000 80000 160000 640000").split()]
tiles = os.environ.get("TILES", "auto").split()
REP = int(os.environ.get("REP", "30"))
print(f"{'rays':>8} {'rows from':>9} {'tile_w':>6} {'median ms':>10} {'min ms':>8} {'max ms':>8} {'Mray/s':>8}  bit-identical to the full-frame launch")
for n in SIZES:
    for r0 in ((0, 350) if n < 640000 else (0,)):
        first = r0 * 800
        rays = full[first:first + n].contiguous()
        for tw in tiles:
            knob = {} if tw == "auto" else {"tile_w": int(tw)}
            with _lib.knobs(**knob):
                out = None
                for _ in range(3): out = f(rays, N_samples=192, **kw)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(REP)]
                for a, b in ev:
                    a.record(); out = f(rays, N_samples=192, **kw); b.record()
                torch.cuda.synchronize()
            ms = np.array([a.elapsed_time(b) for a, b in ev])
            same = bool(torch.equal(out["rgb_map"], ref["rgb_map"][first:first + n]) and torch.equal(out["depth_map"], ref["depth_map"][first:first + n]))
            print(f"{n:>8} {r0:>9} {tw:>6} {np.median(ms):>10.4f} {ms.min():>8.4f} {ms.max():>8.4f} {n / np.median(ms) / 1e3:>8.1f}  {same}")
