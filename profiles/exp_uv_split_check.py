#!/usr/bin/env python3
"""Per-sample density of the UV kernels (fp32 and NGF_UV_F_SPLIT_BF16) against the C oracle on the golden case, for the library named by NGF_LIB
(experiment builds: make -C neural-gauge-fields_amd/csrc expuv NAME=... DEFS=...).  Checker script: imports the oracle like the tests do."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ngf_amd  # noqa: F401
from ngf_amd import _lib, uvmapping
from helpers import load_uv_case
from oracle.oracle import OracleUV

if os.environ.get("NGF_LIB"):
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
for name in ("uv_sphere", "uv_square"):
    g, params = load_uv_case(name)
    pt = str(g["primitive_type"])
    o_color, o_trans, dbg = OracleUV(params, pt).render(g["campos"], g["raydir"], g["U"], bg=g["bg"], debug=True)
    valid = dbg["valid"].astype(bool)
    args = (torch.from_numpy(g["campos"])[None], torch.from_numpy(g["raydir"])[None], torch.from_numpy(g["bg"])[None])
    for split in (False, True):
        m = uvmapping.NeuTex(primitive_type=pt, sample_num=int(g["S"]), device="cuda", split_bf16=split)
        m.load_params(params)
        out = m(*args, jitter_u=torch.from_numpy(g["U"])[None], debug=True)
        sigma, col = out["sigma"][0].cpu().numpy(), out["color"][0].cpu().numpy()
        rel = np.abs(sigma[valid] - dbg["sigma"][valid]) / (np.abs(dbg["sigma"][valid]) + 1e-6)
        bad = rel > 5e-4
        print(f"{os.path.basename(os.path.dirname(_lib.SO_PATH)):12s} {name:9s} split={int(split)}: sigma max rel {rel.max():.2e}, {int(bad.sum())} of {bad.size} beyond 5e-4; pixels max|hip-oracle| {np.abs(col - o_color).max():.2e}"
              + (f"; first bad (ray, sample) {np.argwhere(valid)[np.flatnonzero(bad)[:6]].tolist()}" if bad.any() else ""))
        m.release()
