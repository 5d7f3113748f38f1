"""Same inputs, same handle, thousands of launches: every output bit must repeat.

Round 2 shipped an InfoInv NGF_F_SPLIT_BF16 colour pass whose layer-1 accumulators depended on timing (one 4-ray tile off by ~1e-4 in
about one launch of 50 000: the "failed once in forty suite runs" of test_infoinv_split_bf16_keeps_fp32_accuracy).  Round 3 traced it to the
compiler's packed-math code for the positional-factor chain and replaced it (csrc/ngf_infoinv.hpp `pe_octave`; evidence and the
amplified reproduction in profiles/r03_determinism.txt, profiles/exp_determinism_*.py).  A parity test cannot see such a bug -- the
wrong answer is within tolerance most of the time -- so this file pins the property itself for every kernel family, cheaply
(a launch of these cases takes ~0.1 ms)."""
import numpy as np
import pytest
import torch

from helpers import field_for_case, load_case, load_uv_case

pytestmark = pytest.mark.gpu
LAUNCHES = 4000
# The kernels with bf16 matrix instructions are the family the round-2 defect lived in (1 launch in 4 200): 4 000 launches would catch that
# rate 6 times in 10 -- 50 000 launches miss it with probability e^-12.  The comparison stays on the device (no sync per launch).
LAUNCHES_BF16 = 50000


@pytest.mark.parametrize("name,kw,flags", [
    ("infoinv_r1_on", {"infoinv": True}, {"split_bf16": True}),
    ("infoinv_r1_on", {"infoinv": True}, {}),
    ("infoinv_r1_off", {"infoinv": False}, {"split_bf16": True}),
    ("infoinv_r1_mask", {"infoinv": True}, {"split_bf16": True}),
    ("triplane_r1_gauge", {"iteration": 30001}, {}),
    ("triplane_r1_gauge", {"iteration": 30001}, {"split_bf16": True}),
    ("triplane_r1_mask", {"iteration": 30001}, {"bake": True, "split_bf16": True}),
    ("triplane_r1_mask", {"iteration": 30001}, {"bake": True, "bake_color": True}),
    ("triplane_r1_gauge", {"iteration": 30001}, {"bake": True, "bake_color": True, "split_bf16": True}),        # round 5: level 3 with layer 2 on the bf16 pipe
])
def test_render_repeats_bit_for_bit(name, kw, flags):
    g, params, step, mask = load_case(name)
    S = int(g["S"])
    rays = torch.from_numpy(g["rays"]).cuda()
    f = field_for_case(g, params, mask, **flags)
    first = f(rays, N_samples=S, white_bg=True, **kw)
    rgb0, d0 = first["rgb_map"].clone(), first["depth_map"].clone()
    n = LAUNCHES_BF16 if flags.get("split_bf16") else (20000 if flags.get("bake_color") else LAUNCHES)      # level 3 (the module default): EXEC-masked collect, lane-permuted gather
    from ngf_amd._lib import knobs
    # launch shapes: the library's own plan (a launch this small is all one-ray tiles, one wave per tile); 4-ray tiles with the waves of few
    # workgroups sharing their CU's matrix pipe -- the shape round 2's defect showed in; 8-ray tiles of a frame's bulk on two CUs
    shapes = ({}, {"tile_w": 4, "grid": 4}, {"tile_w": 8, "grid": 2}) if flags.get("split_bf16") else ({}, {"tile_w": 4, "grid": 4})
    total = 0
    for shape in shapes:
        with knobs(**shape):
            moved = torch.zeros((), dtype=torch.int64, device="cuda")
            rgb, depth = torch.empty_like(rgb0), torch.empty_like(d0)
            for _ in range(n):
                f(rays, N_samples=S, white_bg=True, out=(rgb, depth), **kw)
                moved += ((rgb != rgb0).any() | (depth != d0).any()).to(torch.int64)        # bitwise for finite outputs; no host sync in the loop
            moved = int(moved.item())
        assert moved == 0, f"{moved} of {n} launches differ from the first one (launch shape {shape})"
        total += n
    f.release()
    assert torch.isfinite(rgb0).all() and torch.isfinite(d0).all()


def test_uv_render_repeats_bit_for_bit():
    from ngf_amd import uvmapping
    g, params = load_uv_case("uv_sphere")
    for split in (False, True):
        m = uvmapping.NeuTex(primitive_type="sphere", sample_num=int(g["S"]), device="cuda", split_bf16=split)
        m.load_params(params)
        args = (torch.from_numpy(g["campos"])[None].cuda(), torch.from_numpy(g["raydir"])[None].cuda(), torch.from_numpy(g["bg"])[None].cuda())
        U = torch.from_numpy(g["U"])[None].cuda()
        first = m(*args, jitter_u=U)["color"].clone()
        n = LAUNCHES_BF16 if split else 20000          # fp32: round 4 gave the kernel cross-lane LDS writes (positional encodings) and single-tile passes
        moved = torch.zeros((), dtype=torch.int64, device="cuda")
        for _ in range(n):
            moved += (m(*args, jitter_u=U)["color"] != first).any().to(torch.int64)
        moved = int(moved.item())
        m.release()
        assert torch.isfinite(first).all()
        assert moved == 0, (split, moved, n)
