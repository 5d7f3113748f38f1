"""UV-Mapping kernels, 150 000 launches per golden case and precision: every pixel must repeat bit for bit (round 4 changed the kernel's LDS traffic: cross-lane writes of the positional encodings, single-tile passes).  Checker-side script (loads the golden cases through tests/helpers)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, ngf_amd
from ngf_amd import uvmapping, _lib
if os.environ.get("NGF_LIB"):          # an experiment build (make expuv) instead of the product library
    _lib.SO_PATH = os.path.abspath(os.environ["NGF_LIB"])
from helpers import load_uv_case
for name in ("uv_sphere", "uv_square"):
    g, params = load_uv_case(name)
    for split in (False, True):
        m = uvmapping.NeuTex(primitive_type=str(g["primitive_type"]), sample_num=int(g["S"]), device="cuda", split_bf16=split)
        m.load_params(params)
        args = (torch.from_numpy(g["campos"])[None].cuda(), torch.from_numpy(g["raydir"])[None].cuda(), torch.from_numpy(g["bg"])[None].cuda())
        U = torch.from_numpy(g["U"])[None].cuda()
        first = m(*args, jitter_u=U)["color"].clone()
        moved = torch.zeros((), dtype=torch.int64, device="cuda")
        n = int(os.environ.get("N", "150000"))
        for _ in range(n):
            moved += (m(*args, jitter_u=U)["color"] != first).any().to(torch.int64)
        print(name, "split" if split else "fp32", n, "launches,", int(moved.item()), "differ", flush=True)
        m.release()
