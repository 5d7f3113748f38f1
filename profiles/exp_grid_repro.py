"""Does a wave's SECOND tile differ from its first?  infoinv_r1_on is 128 rays = 32 tiles of 4 rays for 32 waves (4 workgroups of 8):
nearly always one tile per wave.  With the "grid" knob the same launch runs on 1 / 2 / 3 workgroups, so every wave takes several tiles;
all outputs must equal the default launch bitwise.   python profiles/exp_grid_repro.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ngf_amd  # noqa
from ngf_amd import _lib
if os.environ.get("NGF_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
from helpers import field_for_case, load_case
for name, kw in (("infoinv_r1_on", {"infoinv": True}), ("infoinv_r1_off", {"infoinv": False}), ("triplane_r1_gauge", {"iteration": 30001})):
    g, params, step, mask = load_case(name)
    S = int(g["S"])
    rays = torch.from_numpy(g["rays"]).cuda()
    for split in (True, False):
        f = field_for_case(g, params, mask, split_bf16=split)
        ref = f(rays, N_samples=S, white_bg=True, **kw)
        for grid in (1, 2, 3):
            for tw in (-1, 8, 16):
                if tw == 16 and name.startswith("triplane"):
                    continue
                with _lib.knobs(grid=grid, tile_w=tw):
                    moved = {}
                    for rep in range(50):
                        r = f(rays, N_samples=S, white_bg=True, **kw)
                        if not torch.equal(r["rgb_map"], ref["rgb_map"]) or not torch.equal(r["depth_map"], ref["depth_map"]):
                            rows = tuple(torch.nonzero((r["rgb_map"] != ref["rgb_map"]).any(1)).flatten().tolist())
                            moved[rows] = moved.get(rows, 0) + 1
                    print(f"{name} split={split} grid={grid} tile_w={tw}: {'identical' if not moved else moved}", flush=True)
        f.release()
