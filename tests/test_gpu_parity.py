"""Parity of the HIP path (through the C ABI, include/ngf.h) against the CPU oracle and the golden
vectors captured from the reference.  Tolerance (BASELINE.json north_star): 1e-4 relative, fp32,
written below as rtol=1e-4 with atol=1e-5 on pixel values in [0,1]; we also assert the much tighter
bound the implementation actually achieves so regressions show up.
"""
import os

import numpy as np
import pytest

from helpers import GOLDEN, big_case, field_for_case, load_case, oracle_for_case, psnr

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5
TRIPLANE = ["triplane_r1_gauge", "triplane_r2_nogauge", "triplane_r1_mask", "triplane_r0"]
INFOINV = ["infoinv_r1_on", "infoinv_r1_off", "infoinv_r1_mask"]           # _mask: InfoInv WITH an alpha mask, black background (round 4)
TRAIN = ["triplane_r1_train_white", "triplane_r1_train_black"]             # reference forwards with is_train=True: supplied jitter + background coin


def _mode(g):
    if "gauge_on" in g:
        return {"iteration": 30001 if int(g["gauge_on"]) else -1}
    return {"infoinv": bool(int(g["infoinv"]))}


def _close(a, b, what, rtol=RTOL, atol=ATOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.size} outside tolerance, max abs {err.max():.3e}"
    return float(err.max())


@pytest.mark.parametrize("bake_color", [False, True])
@pytest.mark.parametrize("name", TRIPLANE + INFOINV)
def test_decode_rgb_matches_oracle(name, bake_color):
    g, params, step, mask = load_case(name)
    if bake_color and str(g["model"]) != "triplane":
        pytest.skip("baked colour is a TriPlane option")
    orc = oracle_for_case(g, params, step, mask)
    from ngf_amd import synth
    n = 1000
    coords = (synth.hash_uniform(77, 1, (n, 6)) * np.float32(2.3) - np.float32(1.15)).astype(np.float32)
    dirs = synth.hash_normal(77, 2, (n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs[:3] = np.eye(3, dtype=np.float32)            # zero components
    f = field_for_case(g, params, mask, bake_color=bake_color)
    mode = int(g["gauge_on"]) if "gauge_on" in g else int(g["infoinv"])
    got = f.decode_rgb(torch.from_numpy(coords), torch.from_numpy(dirs), mode=mode).cpu().numpy()
    want = orc.color_at(coords, dirs)
    assert _close(got, want, "decode_rgb") < 5e-6


@pytest.mark.parametrize("bake", [False, True])
@pytest.mark.parametrize("name", TRIPLANE + INFOINV)
def test_march_matches_oracle(name, bake):
    g, params, step, mask = load_case(name)
    if bake and str(g["model"]) != "triplane":
        pytest.skip("baked density is a TriPlane option")
    orc = oracle_for_case(g, params, step, mask)
    S = int(g["S"])
    _, _, dbg = orc.render(g["rays"], S, debug_rays=g["rays"].shape[0])
    f = field_for_case(g, params, mask, bake=bake)
    mode = int(g["gauge_on"]) if "gauge_on" in g else int(g["infoinv"])
    sigma, weight = f.march(torch.from_numpy(g["rays"]), S, mode=mode)
    sigma, weight = sigma.cpu().numpy(), weight.cpu().numpy()
    # in-box mask must agree exactly: sigma == 0 exactly where the oracle says invalid
    assert np.array_equal(sigma == 0, dbg["sigma"] == 0)
    np.testing.assert_allclose(sigma, dbg["sigma"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(weight, dbg["weight"], rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize("bake", [0, 1, 2, 3])
@pytest.mark.parametrize("name", TRIPLANE + INFOINV + TRAIN)
def test_render_matches_oracle_and_reference(name, bake):
    """bake: bit 0 = NGF_F_BAKE_DENSITY, bit 1 = NGF_F_BAKE_COLOR.  The TRAIN cases are forwards of the reference with is_train=True
    (FieldBase.py:128-130: per-ray jitter; :299: random background): the field gets the captured jitter and coin."""
    g, params, step, mask = load_case(name)
    if bake and str(g["model"]) != "triplane":
        pytest.skip("baked planes are a TriPlane option")
    orc = oracle_for_case(g, params, step, mask)
    S, wb = int(g["S"]), bool(int(g["white_bg"]))
    train_kw, jitter, eff_white = {}, None, wb
    if "is_train" in g:
        jitter = g["jitter"]
        eff_white = wb or float(g["coin"]) < 0.5
        train_kw = {"jitter": torch.from_numpy(jitter), "coin": float(g["coin"])}
    o_rgb, o_depth = orc.render(g["rays"], S, white_bg=eff_white, jitter=jitter)
    f = field_for_case(g, params, mask, bake=bool(bake & 1), bake_color=bool(bake & 2))
    with torch.no_grad():           # the fused launch; with autograd on, is_train=True goes through the differentiable path (tests/test_gpu_autograd.py)
        out = f(torch.from_numpy(g["rays"]).cuda(), white_bg=wb, is_train="is_train" in g, N_samples=S, collect_stats=True, **_mode(g), **train_kw)
    rgb, depth = out["rgb_map"].cpu().numpy(), out["depth_map"].cpu().numpy()
    e1 = _close(rgb, o_rgb, "rgb vs oracle")
    _close(depth, o_depth, "depth vs oracle", atol=5e-5)
    e2 = _close(rgb, g["rgb_map"], "rgb vs reference golden")
    _close(depth, g["depth_map"], "depth vs reference golden", atol=5e-5)
    assert max(e1, e2) < 2e-5 and psnr(rgb, g["rgb_map"]) > 90
    st = f.last_stats.cpu().numpy()
    assert st[3] == g["rays"].shape[0] and st[1] <= st[0] <= g["rays"].shape[0] * S


@pytest.mark.parametrize("level", ["level1", "level2", "no_fold", "split_bf16"])
@pytest.mark.parametrize("name", TRIPLANE + INFOINV + TRAIN)
def test_per_sample_colours_match_the_reference(name, level):
    """SURVEY C2's last intermediates: the reference's own rgb_mask (weight > thr, FieldBase.py:289) and per-sample colours -- outputs of ITS
    compute_rgb / rgb_decoder (Field.py:93-105, networks.py:25-32) at the active samples of the first 8 rays -- against the HIP march's weights
    and the HIP colour stage (ngf_field_decode_rgb) at the reference's own gauge-shifted coordinates: compute_rgb is pinned to a
    reference-held vector, not only through composited pixels."""
    g, params, step, mask = load_case(name)
    tri = str(g["model"]) == "triplane"
    if level in ("level2", "no_fold") and not tri:
        pytest.skip("TriPlane levels")
    flags = {"level1": {}, "level2": {"bake": True}, "no_fold": {"no_fold": True}, "split_bf16": {"split_bf16": True}}[level]
    f = field_for_case(g, params, mask, **flags)
    mode = int(g["gauge_on"]) if "gauge_on" in g else int(g["infoinv"])
    m = g["i_rgb_mask"]
    S = int(g["S"])
    # the march's weights of the same 8 rays (eval-mode cases: the debug march takes no jitter)
    if "is_train" not in g:
        _, weight = f.march(torch.from_numpy(g["rays"][:8]), S, mode=mode)
        weight = weight.cpu().numpy()
        near = np.abs(g["i_weight"] - np.float32(g["thr"])) < 2e-7              # within rounding of the threshold: either side is right
        assert np.array_equal((weight > np.float32(g["thr"]))[~near], m[~near])
        np.testing.assert_allclose(weight, g["i_weight"], rtol=1e-4, atol=2e-7)
    if not m.any():
        assert name == "triplane_r0" and not g["i_rgb"].any()               # the literal random-init preset has no active sample
        return
    dirs = np.ascontiguousarray(np.broadcast_to(g["rays"][:8, None, 3:6], (*m.shape, 3))[m])
    coords = np.ascontiguousarray(g["i_coords"][m])
    got = f.decode_rgb(torch.from_numpy(coords), torch.from_numpy(dirs), mode=mode).cpu().numpy()
    assert np.abs(got - g["i_rgb"][m]).max() < 5e-6, np.abs(got - g["i_rgb"][m]).max()
    assert int(m.sum()) >= 20


def test_ragged_and_tiny_batches():
    g, params, step, mask = load_case("triplane_r1_gauge")
    orc = oracle_for_case(g, params, step, mask)
    f = field_for_case(g, params, mask)
    for n in (1, 2, 63, 64, 65, 129):
        rays = g["rays"][:n]
        o_rgb, o_depth = orc.render(rays, 48)
        out = f(torch.from_numpy(rays).cuda(), N_samples=48, iteration=30001)
        _close(out["rgb_map"].cpu().numpy(), o_rgb, f"rgb n={n}")
        _close(out["depth_map"].cpu().numpy(), o_depth, f"depth n={n}", atol=5e-5)
    out = f(torch.zeros((0, 6)).cuda(), N_samples=48, iteration=30001)
    assert out["rgb_map"].shape == (0, 3) and out["depth_map"].shape == (0,)
    with pytest.raises(ValueError):
        f(torch.zeros((4, 5)).cuda(), N_samples=8)


def test_deterministic_and_chunk_independent():
    g, params, step, mask = load_case("triplane_r2_nogauge")
    f = field_for_case(g, params, mask)
    rays = torch.from_numpy(g["rays"]).cuda()
    a = f(rays, N_samples=40, iteration=30001)
    b = f(rays, N_samples=40, iteration=30001)
    assert torch.equal(a["rgb_map"], b["rgb_map"]) and torch.equal(a["depth_map"], b["depth_map"])
    # the same rays in a different batch composition give bit-identical pixels (no cross-ray coupling)
    c = f(rays[37:150], N_samples=40, iteration=30001)
    assert torch.equal(a["rgb_map"][37:150], c["rgb_map"])


@pytest.mark.parametrize("name", ["triplane_r1_gauge", "triplane_r1_mask", "infoinv_r1_on"])
def test_split_march_is_bit_identical(name):
    """The split march (tile_w rays x 64/tile_w lanes per ray on consecutive steps, csrc/ngf_render.hpp render_kernel<P, true>)
    chains transmittance / acc / depth from lane to lane in step order, so every tile shape gives the pixels of the
    one-ray-per-lane march bit for bit.  S = 45 is not a multiple of any lanes-per-ray count; the batch is ragged."""
    g, params, step, mask = load_case(name)
    f = field_for_case(g, params, mask)
    rays = torch.from_numpy(g["rays"]).cuda()[:203]
    kw = {"infoinv": True} if name.startswith("infoinv") else {"iteration": 30001}
    from ngf_amd._lib import knobs
    with knobs(tile_w=64, split=0, kernel=0):
        ref = f(rays, N_samples=45, **kw)
    for tw in (32, 16, 8, 4, 2, 1):            # 2 / 1: a ray takes two / four 16-lane rows (split_chain_rows)
        with knobs(tile_w=tw, split=1, kernel=0):
            got = f(rays, N_samples=45, **kw)
        assert torch.equal(ref["rgb_map"], got["rgb_map"]), tw
        assert torch.equal(ref["depth_map"], got["depth_map"]), tw
    got = f(rays, N_samples=45, **kw)          # the default choice: 203 rays are all one-ray tiles
    assert torch.equal(ref["rgb_map"], got["rgb_map"]) and torch.equal(ref["depth_map"], got["depth_map"])
    # mixed tile plans (wide tiles first, narrow tiles for the last rays): few workgroups and a small tail so that every width occurs
    many = torch.from_numpy(np.concatenate([g["rays"]] * 8)).cuda()[:1999]
    with knobs(tile_w=64, split=0, kernel=0):
        ref = f(many, N_samples=45, **kw)
    for grid, tail in ((2, 16), (3, 8), (1, 40), (2, 0)):
        with knobs(grid=grid, tail=tail):
            got = f(many, N_samples=45, **kw)
        assert torch.equal(ref["rgb_map"], got["rgb_map"]) and torch.equal(ref["depth_map"], got["depth_map"]), (grid, tail)


@pytest.mark.parametrize("name", INFOINV)
def test_infoinv_split_bf16_keeps_fp32_accuracy(name):
    """NGF_F_SPLIT_BF16 on the InfoInv tree: rgb_decoder (216 features + view -> 64 -> 64 -> 3, csrc/ngf_infoinv.hpp
    mlp_pass16_bf16_ii) AND the density MLP of the march (72 -> 32 -> 32 -> 1, infoinv_sigma_bf16) as 3-term split bf16 products with
    fp32 accumulation.  Same tolerances against oracle and reference goldens, within fp32 rounding noise of the fp32-MFMA path."""
    g, params, step, mask = load_case(name)
    orc = oracle_for_case(g, params, step, mask)
    fs = field_for_case(g, params, mask, split_bf16=True)
    fd = field_for_case(g, params, mask)
    rays = torch.from_numpy(g["rays"]).cuda()
    S = int(g["S"])
    kw = _mode(g)
    wb = bool(int(g["white_bg"]))
    a = fs(rays, N_samples=S, white_bg=wb, **kw)
    b = fd(rays, N_samples=S, white_bg=wb, **kw)
    o_rgb, o_depth = orc.render(g["rays"], S, white_bg=wb)
    ea = _close(a["rgb_map"].cpu().numpy(), o_rgb, "split-bf16 rgb vs oracle")
    _close(a["rgb_map"].cpu().numpy(), g["rgb_map"], "split-bf16 rgb vs reference golden")
    assert ea < 5e-6
    _close(a["depth_map"].cpu().numpy(), o_depth, "split-bf16 depth vs oracle")
    # sigma now comes from split products too: depth and colour sit within fp32 rounding noise of the fp32 path instead of on it
    assert float((a["depth_map"] - b["depth_map"]).abs().max()) < 2e-5 * float(b["depth_map"].abs().max())
    assert float((a["rgb_map"] - b["rgb_map"]).abs().max()) < 5e-6
    from ngf_amd import synth
    n = 203
    coords = (synth.hash_uniform(79, 1, (n, 6)) * np.float32(2.2) - np.float32(1.1)).astype(np.float32)
    coords[:, 2] = coords[:, 1]; coords[:, 4] = coords[:, 0]; coords[:, 5] = coords[:, 3]        # InfoInv's identity split: (x,y),(y,z),(x,z)
    dirs = synth.hash_normal(79, 2, (n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    mode = int(bool(int(g["infoinv"])))
    got = fs.decode_rgb(torch.from_numpy(coords), torch.from_numpy(dirs), mode=mode).cpu().numpy()
    ref = fd.decode_rgb(torch.from_numpy(coords), torch.from_numpy(dirs), mode=mode).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-6
    # the alpha-mask helpers run the split density MLP: fp32-level agreement with the fp32 one
    pts = torch.from_numpy((synth.hash_uniform(79, 3, (64, 3)) * np.float32(3.0) - np.float32(1.5)).astype(np.float32)).cuda()
    np.testing.assert_allclose(fs.compute_alpha(pts, 0.3).cpu().numpy(), fd.compute_alpha(pts, 0.3).cpu().numpy(), rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["triplane_r1_gauge", "triplane_r2_nogauge", "triplane_r1_mask", "triplane_r1_train_white"])
def test_level3_with_layer2_on_the_bf16_pipe_keeps_fp32_accuracy(name):
    """NGF_F_BAKE_DENSITY | NGF_F_BAKE_COLOR | NGF_F_SPLIT_BF16 (round 5, opt-in): level 3 with layer 2 -- all the matrix work level 3 has left --
    as six bf16 products per fp32 product (csrc/ngf_shade_bf16.hpp mlp_pass16_baked_bf16).  Same tolerances against the oracle and the
    reference's golden pixels as the fp32 path, within fp32 rounding noise of level 3 itself, the march untouched (depth bit-identical),
    every launch shape the same bits."""
    g, params, step, mask = load_case(name)
    orc = oracle_for_case(g, params, step, mask)
    fs = field_for_case(g, params, mask, bake=True, bake_color=True, split_bf16=True)
    fd = field_for_case(g, params, mask, bake=True, bake_color=True)
    rays = torch.from_numpy(g["rays"]).cuda()
    S, white = int(g["S"]), bool(int(g["white_bg"]))
    kw = _mode(g)
    train_kw, jitter, eff_white = {}, None, white
    if "is_train" in g:
        jitter = g["jitter"]
        eff_white = white or float(g["coin"]) < 0.5
        train_kw = {"jitter": torch.from_numpy(jitter), "coin": float(g["coin"]), "is_train": True}
    with torch.no_grad():
        a = fs(rays, N_samples=S, white_bg=white, **kw, **train_kw)
        b = fd(rays, N_samples=S, white_bg=white, **kw, **train_kw)
    o_rgb, o_depth = orc.render(g["rays"], S, white_bg=eff_white, jitter=jitter)
    ea = _close(a["rgb_map"].cpu().numpy(), o_rgb, "level 3 + bf16 layer 2: rgb vs oracle")
    _close(a["rgb_map"].cpu().numpy(), g["rgb_map"], "level 3 + bf16 layer 2: rgb vs reference golden")
    _close(a["depth_map"].cpu().numpy(), g["depth_map"], "level 3 + bf16 layer 2: depth vs reference golden", atol=5e-5)
    assert ea < 5e-6
    assert torch.equal(a["depth_map"], b["depth_map"])                                   # the march is the same code
    assert float((a["rgb_map"] - b["rgb_map"]).abs().max()) < 2e-6
    from ngf_amd._lib import knobs
    if "is_train" not in g:
        for n in (1, 7, 65):                                                             # ragged launches
            c = fs(rays[:n], N_samples=S, white_bg=white, **kw)
            assert torch.equal(c["rgb_map"], a["rgb_map"][:n])
        for tw in (1, 2, 4, 8, 64):                                                      # every tile shape, the unsplit 64-ray tile included (view-input MFMAs instead of the per-ray fold)
            with knobs(tile_w=tw):
                c = fs(rays, N_samples=S, white_bg=white, **kw)
            if tw <= 8:
                assert torch.equal(c["rgb_map"], a["rgb_map"]) and torch.equal(c["depth_map"], a["depth_map"]), tw
            else:
                assert float((c["rgb_map"] - a["rgb_map"]).abs().max()) < 2e-6, tw
    # the colour stage alone, per sample
    from ngf_amd import synth
    n = 203
    coords = (synth.hash_uniform(78, 1, (n, 6)) * np.float32(2.2) - np.float32(1.1)).astype(np.float32)
    dirs = synth.hash_normal(78, 2, (n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    got = fs.decode_rgb(torch.from_numpy(coords), torch.from_numpy(dirs), mode=1).cpu().numpy()
    assert np.abs(got - orc.color_at(coords, dirs)).max() < 5e-6
    fs.release(); fd.release()


@pytest.mark.parametrize("bake_density", [False, True])
@pytest.mark.parametrize("name", ["triplane_r1_gauge", "triplane_r2_nogauge", "triplane_r1_mask"])
def test_split_bf16_colour_mlp_keeps_fp32_accuracy(name, bake_density):
    """NGF_F_SPLIT_BF16 (opt-in): the colour MLP's products on the bf16 matrix pipe with 3-term split operands and fp32 accumulation
    (csrc/ngf_shade_bf16.hpp).  Same tolerance against the oracle and the reference's golden pixels as the fp32-MFMA path, and
    within fp32 rounding noise of that path (the dropped cross terms are < 2^-24 of a product)."""
    g, params, step, mask = load_case(name)
    orc = oracle_for_case(g, params, step, mask)
    fs = field_for_case(g, params, mask, bake=bake_density, split_bf16=True)
    fd = field_for_case(g, params, mask, bake=bake_density)
    rays = torch.from_numpy(g["rays"]).cuda()
    S = int(g["S"])
    white = bool(int(g["white_bg"]))
    kw = _mode(g)
    a = fs(rays, N_samples=S, white_bg=white, **kw)
    b = fd(rays, N_samples=S, white_bg=white, **kw)
    o_rgb, o_depth = orc.render(g["rays"], S, white_bg=white)
    ea = _close(a["rgb_map"].cpu().numpy(), o_rgb, "split-bf16 rgb vs oracle")
    _close(a["rgb_map"].cpu().numpy(), g["rgb_map"], "split-bf16 rgb vs reference golden")
    assert ea < 5e-6
    assert torch.equal(a["depth_map"], b["depth_map"])                                   # the march is the same code
    assert float((a["rgb_map"] - b["rgb_map"]).abs().max()) < 2e-6
    for n in (1, 7, 65):                                                                 # ragged launches
        c = fs(rays[:n], N_samples=S, white_bg=white, **kw)
        assert torch.equal(c["rgb_map"], a["rgb_map"][:n])
    # the 12-waves-per-CU form of the pass (taps in two rows, fragments two tiles at a time: mlp_pass16_bf16_rows) is the same arithmetic
    # (an experiment kernel: libngf_hip_exp.so, its own handle)
    from ngf_amd import _lib
    with _lib.library("exp"):
        fs12 = field_for_case(g, params, mask, bake=bake_density, split_bf16=True)
        with _lib.knobs(waves=12):
            a12 = fs12(rays, N_samples=S, white_bg=white, **kw)
        fs12.release()
    assert torch.equal(a12["rgb_map"], a["rgb_map"]) and torch.equal(a12["depth_map"], a["depth_map"])
    # the colour stage alone, per sample
    from ngf_amd import synth
    n = 203
    coords = (synth.hash_uniform(78, 1, (n, 6)) * np.float32(2.2) - np.float32(1.1)).astype(np.float32)
    dirs = synth.hash_normal(78, 2, (n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    got = fs.decode_rgb(torch.from_numpy(coords), torch.from_numpy(dirs), mode=1).cpu().numpy()
    assert np.abs(got - orc.color_at(coords, dirs)).max() < 5e-6


@pytest.mark.parametrize("name", ["triplane_r1_gauge", "triplane_r2_nogauge", "triplane_r1_mask"])
def test_no_fold_level0_matches_reference(name):
    """NGF_F_NO_FOLD: rgb_decoder exactly as written (networks.py:25-30) -- `basis` as its own 144x144 matrix stage, view inputs
    per sample -- against the oracle, the reference's golden pixels, and the default (pre-composed layer 1, per-ray view fold)
    path, which may differ from it by rounding only."""
    g, params, step, mask = load_case(name)
    orc = oracle_for_case(g, params, step, mask)
    f0 = field_for_case(g, params, mask, no_fold=True)
    f1 = field_for_case(g, params, mask)
    rays = torch.from_numpy(g["rays"]).cuda()
    S = int(g["S"])
    white = bool(int(g["white_bg"]))
    kw = _mode(g)
    a = f0(rays, N_samples=S, white_bg=white, **kw)
    b = f1(rays, N_samples=S, white_bg=white, **kw)
    o_rgb, o_depth = orc.render(g["rays"], S, white_bg=white)
    _close(a["rgb_map"].cpu().numpy(), o_rgb, "no-fold rgb vs oracle")
    _close(a["rgb_map"].cpu().numpy(), g["rgb_map"], "no-fold rgb vs reference golden")
    assert torch.equal(a["depth_map"], b["depth_map"])                                   # the march is the same code
    assert float((a["rgb_map"] - b["rgb_map"]).abs().max()) < 1e-5                       # folds (i) + (ii): rounding only
    # the colour stage alone
    from ngf_amd import synth
    n = 203
    coords = (synth.hash_uniform(77, 1, (n, 6)) * np.float32(2.2) - np.float32(1.1)).astype(np.float32)
    dirs = synth.hash_normal(77, 2, (n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    got = f0.decode_rgb(torch.from_numpy(coords), torch.from_numpy(dirs), mode=1).cpu().numpy()
    assert np.abs(got - orc.color_at(coords, dirs)).max() < 5e-6
    with pytest.raises((RuntimeError, ValueError)):                                      # level 0 excludes the bakes (constructor and ngf_field_create)
        field_for_case(g, params, mask, no_fold=True, bake=True).handle()
    # the documented level-0 option through the PUBLIC constructor alone: bake_density's default resolves to "not no_fold"
    from ngf_amd import triplane
    fp = triplane.TriPlane(torch.tensor(np.asarray(g["aabb"], np.float32)), [int(v) for v in g["grid"]], "cuda", no_fold=True, gauge_start=0,
                           near_far=[float(v) for v in g["near_far"]], distance_scale=float(g["distance_scale"]),
                           rayMarch_weight_thres=float(g["thr"]), step_ratio=float(g["step_ratio"]))
    assert fp.no_fold and not fp.bake_density
    fp.load_params(params)
    if mask is None:
        c = fp(rays, N_samples=S, white_bg=white, **kw)
        assert torch.equal(c["rgb_map"], a["rgb_map"]) and torch.equal(c["depth_map"], a["depth_map"])
    else:
        fp.handle()


@pytest.mark.parametrize("name", ["triplane_r1_gauge", "triplane_r2_nogauge", "triplane_r1_mask"])
def test_lds_staged_strips_are_bit_identical(name):
    """(runs on libngf_hip_exp.so: the product library carries only kernels that can be the default, csrc/Makefile)"""
    from ngf_amd import _lib
    with _lib.library("exp"):
        _test_lds_staged_strips_are_bit_identical_impl(name)


def _test_lds_staged_strips_are_bit_identical_impl(name):
    """The LDS-staged texture variant (csrc/ngf_stage.hpp, knob stage = 1): gauge strips (gauge on) / density strips (gauge off,
    8 waves per CU) are loaded once per tile iteration into LDS and tapped from there; the arithmetic on the fetched values is
    unchanged, so the pixels are the gather kernel's bit for bit, whether an iteration's rectangle fits the strip or falls back."""
    from ngf_amd._lib import knobs
    g, params, step, mask = load_case(name)
    f = field_for_case(g, params, mask)
    rays = torch.from_numpy(g["rays"]).cuda()
    for it in (30001, -1):                   # gauge on / off
        for white in (True, False):
            ref = f(rays, N_samples=45, white_bg=white, iteration=it)
            for waves in (12, 8):
                for tw in (8, 4):
                    with knobs(stage=1, waves=waves, tile_w=tw, kernel=0):
                        got = f(rays, N_samples=45, white_bg=white, iteration=it, collect_stats=True)
                    assert torch.equal(ref["rgb_map"], got["rgb_map"]) and torch.equal(ref["depth_map"], got["depth_map"]), (it, white, waves, tw)


@pytest.mark.parametrize("bake", [0, 1, 2, 3])
@pytest.mark.parametrize("name", TRIPLANE)
def test_specialised_kernel_is_bit_identical(name, bake):
    """(runs on libngf_hip_exp.so: the product library carries only kernels that can be the default, csrc/Makefile)"""
    from ngf_amd import _lib
    with _lib.library("exp"):
        _test_specialised_kernel_is_bit_identical_impl(name, bake)


def _test_specialised_kernel_is_bit_identical_impl(name, bake):
    """The specialised variant (knob kernel = 1; NOT the default: it is bit-identical and slower) splits the waves of a CU
    into march waves and shade waves (csrc/ngf_render_pc.hpp: LDS record
    queues, ray-major lanes with a DPP chain, per-ray colour sums in record order).  Every ray sees the arithmetic of the fused
    kernel in the same order, so the pixels are the fused kernel's bit for bit -- for both tile widths, ragged batches and
    S = 45 (not a multiple of any lanes-per-ray count)."""
    from ngf_amd._lib import knobs
    g, params, step, mask = load_case(name)
    f = field_for_case(g, params, mask, bake=bool(bake & 1), bake_color=bool(bake & 2))
    kw = _mode(g)
    for n in (1, 7, 203, len(g["rays"])):
        rays = torch.from_numpy(g["rays"]).cuda()[:n]
        for white in (True, False):
            for tw in (8, 4):
                # reference: the fused kernel, one ray per lane; with baked colour planes the small-tile kernels start a sample's
                # layer-1 sum from the per-ray view term (the unsplit one adds it last), so there the reference is the fused
                # kernel at the same tile width
                with knobs(kernel=0, tile_w=tw if bake & 2 else 64, split=1 if bake & 2 else 0):
                    ref = f(rays, N_samples=45, white_bg=white, collect_stats=True, **kw)
                st_ref = f.last_stats.clone()
                with knobs(kernel=1, tile_w=tw):
                    got = f(rays, N_samples=45, white_bg=white, collect_stats=True, **kw)
                assert torch.equal(ref["rgb_map"], got["rgb_map"]), (n, white, tw)
                assert torch.equal(ref["depth_map"], got["depth_map"]), (n, white, tw)
                assert int(st_ref[1]) == int(f.last_stats[1]) and int(f.last_stats[3]) == n      # active samples, rays (the evaluated count depends on the tile shape: early termination is per tile)


def test_specialised_kernel_full_frame_bit_identical():
    """(runs on libngf_hip_exp.so: the product library carries only kernels that can be the default, csrc/Makefile)"""
    from ngf_amd import _lib
    with _lib.library("exp"):
        _test_specialised_kernel_full_frame_bit_identical_impl()


def _test_specialised_kernel_full_frame_bit_identical_impl():
    """The same on the headline frame (640 000 rays, S = 192, R1 and the MLP-stress preset R2) and on an 80 000-ray shard."""
    from ngf_amd._lib import knobs
    from ngf_amd import rays as nrays, synth
    rays = nrays.generate_rays(800, 800, nrays.blender_focal(800), synth.lookat_pose())
    for preset in ("R1", "R2", "R0"):
        g, params, step = big_case("triplane", preset)
        f = field_for_case(g, params, None)
        for sub in (rays, rays[:80000]):
            with knobs(kernel=0):
                ref = f(sub, N_samples=192, iteration=30001)
            with knobs(kernel=1):
                got = f(sub, N_samples=192, iteration=30001)
            assert torch.equal(ref["rgb_map"], got["rgb_map"]) and torch.equal(ref["depth_map"], got["depth_map"]), preset
        f.release()


@pytest.mark.parametrize("name,bias", [("triplane_r1_gauge", None), ("triplane_r1_gauge", 25.0), ("triplane_r1_mask", 14.0), ("infoinv_r1_on", None)])
def test_early_termination_is_bit_identical(name, bias):
    """The march stops a tile once, for all its rays, T is below half an ulp of acc and of depth / z_max (and below the
    colour threshold): no later sample can change an output bit; iterations in which no lane has a valid sample are skipped
    before their gathers.  Checked against the full march (knob ablate = 96 switches both off) on the
    golden scenes, on opaque variants of them (a surface right at the box entry: termination after a few steps) and
    against the oracle, with and without a white background, S = 160."""
    g, params, step, mask = load_case(name)
    if bias is not None:
        params = dict(params)
        params["density_decoder.bias"] = np.array([bias], np.float32)
    f = field_for_case(g, params, mask)
    orc = oracle_for_case(g, params, step, mask)
    rays = torch.from_numpy(g["rays"]).cuda()
    kw = {"infoinv": True} if name.startswith("infoinv") else {"iteration": 30001}
    from ngf_amd._lib import knobs
    for white in (True, False):
        # a launch this small is all one-ray tiles by default (64 steps per iteration: these rays cross the box in fewer): the 8-ray tiles of a
        # frame's bulk (8 steps per iteration) are where the early stop shows; both are checked for identical bits
        for tw in (8, -1):
            with knobs(ablate=96, tile_w=tw):                    # 32: no early termination, 64: no empty-iteration skip
                full = f(rays, N_samples=160, white_bg=white, collect_stats=True, **kw)
            n_full = int(f.last_stats[0])
            with knobs(tile_w=tw):
                early = f(rays, N_samples=160, white_bg=white, collect_stats=True, **kw)
            n_early = int(f.last_stats[0])
            assert torch.equal(full["rgb_map"], early["rgb_map"]) and torch.equal(full["depth_map"], early["depth_map"])
            assert n_early <= n_full
            if bias is not None and tw == 8:
                assert n_early < 0.5 * n_full              # the opaque scenes really stop early
        o_rgb, o_depth = orc.render(g["rays"], 160, white_bg=white)
        np.testing.assert_allclose(early["rgb_map"].cpu().numpy(), o_rgb, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(early["depth_map"].cpu().numpy(), o_depth, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("model,preset", [("triplane", "R1"), ("triplane", "R2"), ("infoinv", "R1")])
def test_headline_geometry_chunk(model, preset):
    """4096 rays x 192 samples of the 800x800 frame on 256^2 planes (BASELINE config 2/3 shapes)."""
    from ngf_amd import synth
    g, params, step = big_case(model, preset)
    frame = synth.lookat_rays(800, 800, rows=(396, 404))       # 8 rows through the image centre
    rays = frame[::2][:3200]
    g["model"] = np.array(model)
    g["gauge_on"] = np.array(1)
    g["infoinv"] = np.array(1)
    orc = oracle_for_case(g, params, step, None)
    o_rgb, o_depth = orc.render(rays, 192)
    f = field_for_case(g, params, None)
    kw = {"iteration": 30001} if model == "triplane" else {"infoinv": True}
    out = f(torch.from_numpy(rays).cuda(), N_samples=192, collect_stats=True, **kw)
    rgb, depth = out["rgb_map"].cpu().numpy(), out["depth_map"].cpu().numpy()
    e = _close(rgb, o_rgb, "rgb")
    _close(depth, o_depth, "depth", atol=5e-5)
    st = f.last_stats.cpu().numpy()
    print(f"{model} {preset}: max abs err {e:.2e}, PSNR {psnr(rgb, o_rgb):.1f} dB, active fraction {st[1] / (3200 * 192):.3f}")
    if model == "triplane":
        for bd, bc in ((True, False), (False, True), (True, True)):
            fb = field_for_case(g, params, None, bake=bd, bake_color=bc)
            outb = fb(torch.from_numpy(rays).cuda(), N_samples=192, iteration=30001)
            eb = _close(outb["rgb_map"].cpu().numpy(), o_rgb, f"rgb (bake_density={bd}, bake_color={bc})")
            print(f"   bake_density={bd} bake_color={bc}: max abs err {eb:.2e}")


def test_generate_rays_matches_reference():
    """ngf_generate_rays vs rays produced by the reference's own get_ray_directions / get_rays with the Blender loader's
    call pattern (tests/golden/rays_blender.npz, make_golden.capture_rays): origins bit-exact, directions to matmul rounding."""
    from ngf_amd import rays as nrays
    fx = np.load(os.path.join(GOLDEN, "rays_blender.npz"))
    for tag in ("a", "b"):
        H, W, c2w = int(fx[tag + "_H"]), int(fx[tag + "_W"]), fx[tag + "_c2w"]
        focal = float(np.float32(fx[tag + "_focal"]))               # torch rounds the loader's double focal to float32 once
        whole = nrays.generate_rays(H, W, focal, c2w).cpu().numpy().reshape(H, W, 6)
        for k, r in enumerate(fx[tag + "_rows"]):
            want = fx[tag + "_rays"][k]
            got = nrays.generate_rays(H, W, focal, c2w, rows=(int(r), int(r) + 1)).cpu().numpy()
            assert np.array_equal(got, whole[int(r)])               # row blocks = rows of the whole frame (sharded render)
            assert np.array_equal(got[:, :3], want[:, :3])
            assert np.abs(got[:, 3:] - want[:, 3:]).max() <= 3e-7
    assert nrays.blender_focal(800) == float(np.float32(fx["a_focal"]))
    # and the renders of the reference's rays and the device's rays agree
    g, params, step = big_case("triplane", "R1")
    f = field_for_case(g, params, None)
    want = fx["a_rays"][3:6].reshape(-1, 6)
    got = nrays.generate_rays(800, 800, nrays.blender_focal(800), fx["a_c2w"], rows=(399, 402))
    a = f(torch.from_numpy(want).cuda(), N_samples=96, iteration=30001)["rgb_map"]
    b = f(got, N_samples=96, iteration=30001)["rgb_map"]
    # a 1-ulp change of a direction can move a sample across the box face or a texel edge (SURVEY.md section 7,
    # hazard 1), so single pixels may differ by ~1e-3; everything else agrees to rounding
    diff = (a - b).abs()
    assert float(diff.max()) < 5e-3 and float((diff > 1e-5).float().mean()) < 1e-2


def test_generate_rays_dtu_matches_reference():
    """ngf_generate_rays_dtu vs the reference's own get_rays_dir on the shipped DTU cameras (tests/golden/rays_dtu.npz)."""
    from ngf_amd import rays as nrays
    from ngf_amd import synth
    fx = np.load(os.path.join(GOLDEN, "rays_dtu.npz"))
    H, W = int(fx["H"]), int(fx["W"])
    for v in (0, 33):
        cam = (fx[f"v{v}_focal"], fx[f"v{v}_princpt"], fx[f"v{v}_rot"])
        for k, r in enumerate(fx[f"v{v}_rows"]):
            got = nrays.generate_rays_dtu(H, W, *cam, rows=(int(r), int(r) + 1)).cpu().numpy()
            want = fx[f"v{v}_raydir"][k]
            assert np.abs(got - want).max() <= 3e-7
            assert np.mean(got != want) < 0.01                      # same float32 operation order: expected bit-exact
    v0 = synth.DTU_VIEW0
    assert np.array_equal(np.float32(v0["focal"]), fx["v0_focal"]) and np.array_equal(np.float32(v0["rot"]), fx["v0_rot"])
    assert np.array_equal(np.float32(v0["princpt"]), fx["v0_princpt"]) and np.array_equal(np.float32(v0["campos"]), fx["v0_campos"])


def test_eight_rank_data_path_on_one_gpu():
    """BASELINE config 5 (800x800 frame sharded 8-way) without a second GPU: the 8 ranks' row sets (10-row blocks dealt round
    robin, ngf_amd.dist.interleaved_rows) are generated on the device and rendered ONE AFTER THE OTHER into the send buffers of
    ngf_amd.dist (shard_buffers: [3*per | per] float32, what all_gather_into_tensor concatenates rank-major), then put back in
    image order by dist.deinterleave: bit-identical to the single-launch frame.  Covers the 80 000-ray / tile_w = 4 launch shape
    every rank runs on the 8-GPU node, the packing of the exchange buffer and the reorder."""
    from ngf_amd import dist as ndist
    from ngf_amd import rays as nrays, synth
    H = W = 800
    world, block = 8, 10
    per = H * W // world
    g, params, step = big_case("triplane", "R1")
    f = field_for_case(g, params, None)
    focal, c2w = nrays.blender_focal(W), synth.lookat_pose()
    whole = f(nrays.generate_rays(H, W, focal, c2w), N_samples=192, white_bg=True, iteration=30001)
    recv = torch.empty((world, 4 * per), device="cuda")                     # rank-major, as the all-gather lays it out
    for rank in range(world):
        rows = ndist.interleaved_rows(H, world, rank, block)
        rays = torch.cat([nrays.generate_rays(H, W, focal, c2w, rows=r) for r in rows], 0)
        assert rays.shape[0] == per
        send, rgb_view, depth_view = ndist.shard_buffers(per, "cuda")
        f(rays, N_samples=192, white_bg=True, iteration=30001, out=(rgb_view, depth_view))
        recv[rank] = send
    rgb = recv[:, : 3 * per].reshape(world * per, 3)
    depth = recv[:, 3 * per:].reshape(world * per)
    rgb, depth = ndist.deinterleave(rgb, depth, H, W, world, block)
    assert torch.equal(rgb, whole["rgb_map"]) and torch.equal(depth, whole["depth_map"])


def _run_bench(extra_args, env_extra, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + extra_args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("{") and len(last) < 4096, (len(last), last[:300])        # the driver keeps an ~8 KB tail: ONE short last line
    assert sum(ln.startswith("{") for ln in r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]
    return json.loads(last), root


def test_bench_rccl_path_single_gpu(tmp_path):
    """The N>1 path of bench.py (nccl init, barrier, all_gather of composited pixels) on ONE GPU; the line it prints is the compact one."""
    d, _ = _run_bench(["--steps", "2", "--warmup", "1", "--extras", "0", "--cpu-seconds", "0"],
                      dict(NGF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert d["n_gpus"] == 1 and d["value"] > 1 and d["scaling"] == "strong"
    assert d["gathered_frame_bit_identical_to_single_gpu_render"] is True
    assert d["all_gather_ms"] > 0 and 0 < d["shard_kernel_ms"]["min"] <= d["shard_kernel_ms"]["max"] and d["roofline"]["frac"] > 0
    cp = d["critical_path_ms"]          # one unpipelined frame: render -> all-gather -> reorder
    assert cp["render"] > 0 and cp["all_gather"] > 0 and cp["reorder"] > 0 and abs(cp["sum"] - cp["render"] - cp["all_gather"] - cp["reorder"]) < 1e-3
    assert d["launch_ms"]["min"] <= d["ms_per_step_median"] <= d["launch_ms"]["max"]


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run
    (the driver's N = 1 command line is launcher-less; an 8-GPU SCALE run started the same way must not die on a usage error).  One GPU can
    only run N = 1, so NGF_BENCH_FORCE_LAUNCH=1 takes the self-launch route at N = 1: parent -> torchrun -> one rank over RCCL -> ONE JSON line."""
    import os
    env = {k: "" for k in ()}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        assert k not in os.environ, "this test must start from a launcher-less environment"
    d, _ = _run_bench(["--steps", "2", "--warmup", "1", "--extras", "0", "--cpu-seconds", "0"], dict(env, NGF_BENCH_FORCE_LAUNCH="1"))
    assert d["config"]["launcher"].startswith("self") and d["n_gpus"] == 1 and d["value"] > 1
    assert d["gathered_frame_bit_identical_to_single_gpu_render"] is True and d["all_gather_ms"] > 0
    assert d["critical_path_ms"]["render"] > 0


@pytest.mark.parametrize("world", [2, 4])
def test_bench_with_real_ranks_on_one_gpu(world):
    """The N > 1 path of bench.py with REAL ranks: `python bench.py --gpus N` (no launcher: it starts its own torch.distributed.run) as N processes that share the
    one GPU of this box -- RCCL refuses two ranks on a device, so the exchange runs over gloo (CUDA tensors through the host; NGF_BENCH_BACKEND / NGF_BENCH_ONE_DEVICE,
    test knobs).  Everything else is the code an 8-GPU node runs: every rank marches ITS interleaved row blocks on two alternating render streams, the
    double-buffered all-gather crosses process boundaries, the ranks reach the timed region at different times (clock preamble behind a barrier, no collective
    inside it), rank 0 prints ONE line -- and the gathered, re-ordered frame is bit-identical to a single-launch render on every rank."""
    import json
    import os
    import subprocess
    import sys
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        assert k not in os.environ, "this test must start from a launcher-less environment"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NGF_BENCH_BACKEND="gloo", NGF_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 6 and d["value"] > 0.1 and d["scaling"] == "strong"
    assert d["config"]["launcher"].startswith("self") and "gloo" in d["config"]["test_backend"]
    assert d["gathered_frame_bit_identical_to_single_gpu_render"] is True
    assert 0 < d["shard_kernel_ms"]["min"] <= d["shard_kernel_ms"]["max"] and d["all_gather_ms"] > 0
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only


def test_bench_default_line_is_parseable_with_extras():
    """The driver's command (`bench.py --gpus 1 --steps K --warmup W`, extras and CPU baseline ON): one final JSON line < 4 KB carrying
    value, roofline and cpu_baseline; the extras live in bench_extras.json (the 29 KB line left the round unmeasured)."""
    import json
    import os
    d, root = _run_bench(["--steps", "3", "--warmup", "1", "--cpu-seconds", "2"], {})
    assert d["value"] > 1 and d["ms_per_step"] > 0 and d["unit"] == "Mray/s" and d["dtype"] == "f32"
    rf = d["roofline"]
    # round 5: `bound` names what binds -- the SIMD's fp32 datapath, which MFMA and VALU instructions share; the executed-MFMA
    # fraction and the useful-op fraction travel next to it (the counters' busy fractions live under `physical`)
    assert rf["bound"] == "simd" and 0 < rf["frac"] <= 1.0 and rf["kernel_ms"] > 0 and 0 < rf["mfma_frac"] < rf["frac"]
    assert 0 < rf.get("useful_op_frac", 0.5) <= 1.0
    # the per-step launch times come from HIP events INSIDE the timed loop; roofline.kernel_ms is their median
    assert d["launch_ms"]["min"] <= d["ms_per_step_median"] <= d["launch_ms"]["max"] and rf["kernel_ms"] == pytest.approx(d["ms_per_step_median"], rel=1e-3)
    assert d["ms_per_step_median"] <= d["ms_per_step"] * 1.02           # the wall clock per step also holds the launch gaps
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # round 6: frac is work / time / peak of THIS run -- executed matrix flops (pass count of this run's statistics launch) plus
    # the necessary vector arithmetic (128 flop per wave64 instruction), over the timed loop's median launch -- not a figure loaded from profiles/
    redo = (rf["flops_per_launch"] + 128.0 * rf["valu_insts_necessary"]) / (rf["kernel_ms"] * 1e-3) / 157.3e12
    assert rf["frac"] == pytest.approx(redo, rel=2e-3) and rf["mfma_frac"] == pytest.approx(rf["flops_per_launch"] / (rf["kernel_ms"] * 1e-3) / 157.3e12, rel=2e-3)
    assert rf["flops_per_launch"] > rf["mlp_passes"] * 64 * 2048 * 0.999 and rf["clock_ghz"] == 2.4
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"]
    assert d["parity"]["max_abs_err_vs_cpu_port"] < 5e-5
    side = json.load(open(os.path.join(root, d["extras_file"])))
    assert side["value"] == pytest.approx(d["value"], rel=1e-4)
    assert not [k for k, v in side["extras"].items() if "error" in v], side["extras"]
    for k in ("triplane_R0", "triplane_R2", "infoinv_R1", "uvmapping_sphere", "train_step_R1", "eval_output_stage_800x800", "triplane_R1_S884_mask",
              "triplane_R2_S884_mask", "triplane_R1_S884_ball", "handle_build_ms"):
        assert k in side["extras"]
    assert set(side["extras"]["handle_build_ms"]) >= {"triplane_level0", "triplane_level3", "infoinv", "uvmapping"} and "handle_build_ms" in d
    assert set(d["extras_Mray_s"]) <= set(side["extras"])


@pytest.mark.parametrize("model,preset", [("triplane", "R1"), ("infoinv", "R1")])
def test_dense_eighth_of_the_frame_against_the_oracle(model, preset):
    """VERDICT r5 item 9 / weak #1a: the suite's full-frame oracle checks are strided (1 ray in 194), the dense one was a one-off script
    (profiles/exp_full_frame_parity.py).  Here EVERY ray of 100 image rows through the object (rows 350-449: 80 000 rays, S = 192) goes through
    the C oracle on the host cores and is compared value by value -- TriPlane at the module's default level 3 and at level 3 with layer 2 on
    the bf16 pipe, InfoInv at fp32.  Tolerance: north_star's 1e-4 relative (+1e-5 absolute on [0,1] pixels); the largest differences are single
    samples whose weight sits within an ulp of the 1e-4 colour threshold (SURVEY 7 hazard 3)."""
    import os
    from ngf_amd import synth
    g, params, step = big_case(model, preset)
    g["gauge_on"] = np.array(1); g["infoinv"] = np.array(1)
    rays_np = synth.lookat_rays(800, 800, rows=(350, 450))
    assert rays_np.shape == (80000, 6)
    rays = torch.from_numpy(rays_np).cuda()
    orc = oracle_for_case(g, params, step, None)
    o_rgb, o_depth = orc.render(rays_np, 192, white_bg=True, threads=min(128, os.cpu_count() or 1))
    tri = model == "triplane"
    kw = {"iteration": 30001} if tri else {"infoinv": True}
    for flags in ([dict(bake=True, bake_color=True), dict(bake=True, bake_color=True, split_bf16=True)] if tri else [dict()]):
        f = field_for_case(g, params, None, **flags)
        out = f(rays, N_samples=192, white_bg=True, **kw)
        rgb, depth = out["rgb_map"].cpu().numpy().astype(np.float64), out["depth_map"].cpu().numpy().astype(np.float64)
        for name, a, b in (("rgb", rgb, o_rgb), ("depth", depth, o_depth)):
            d = np.abs(a - b)
            bad = d > (1e-5 + 1e-4 * np.abs(b))
            mse = float((d ** 2).mean())
            assert int(bad.sum()) == 0, f"{model} {preset} {flags} {name}: {int(bad.sum())} of {bad.size} values beyond rtol 1e-4 + atol 1e-5, max abs {d.max():.3e}"
            assert mse == 0 or -10 * np.log10(mse) > 110.0, (model, flags, name)
        f.release()


@pytest.mark.parametrize("level", [1, 3])
def test_full_frame_properties(level):
    """BASELINE config 2 at full size (800x800, S=192, R1): size-independent properties + a strided oracle check -- at level 1 and at level 3,
    the module's default since round 4 (what bench.py's headline runs)."""
    from ngf_amd import synth
    g, params, step = big_case("triplane", "R1")
    g["gauge_on"] = np.array(1)
    rays_np = synth.lookat_rays(800, 800)
    rays = torch.from_numpy(rays_np).cuda()
    f = field_for_case(g, params, None, bake=level == 3, bake_color=level == 3)
    full = f(rays, N_samples=192, white_bg=True, iteration=30001, collect_stats=True)
    st = f.last_stats.cpu().numpy()
    assert st[3] == 640000 and 0.2 < st[1] / (640000 * 192) < 0.3          # active fraction of preset R1 (0.235)
    rgb, depth = full["rgb_map"], full["depth_map"]
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    # rays are independent: any subset rendered on its own gives bit-identical pixels
    idx = torch.arange(0, 640000, 97, device="cuda")
    sub = f(rays[idx], N_samples=192, white_bg=True, iteration=30001)
    assert torch.equal(sub["rgb_map"], rgb[idx]) and torch.equal(sub["depth_map"], depth[idx])
    # white background only adds (1 - acc) >= 0 before the clamp
    black = f(rays[idx], N_samples=192, white_bg=False, iteration=30001)["rgb_map"]
    assert bool((sub["rgb_map"] + 1e-6 >= black).all())
    delta = (sub["rgb_map"] - black)
    assert float((delta.max(dim=1).values - delta.min(dim=1).values).max()) < 2e-6   # the same (1-acc) on all three channels (no clamp hit at R1)
    # strided oracle check on the same frame: every 194th ray (3 300 rays; round 4: every 776th)
    pick = idx.cpu().numpy()[::2]
    orc = oracle_for_case(g, params, step, None)
    o_rgb, o_depth = orc.render(rays_np[pick], 192)
    _close(rgb[torch.from_numpy(pick).cuda()].cpu().numpy(), o_rgb, "full-frame rgb vs oracle")
    _close(depth[torch.from_numpy(pick).cuda()].cpu().numpy(), o_depth, "full-frame depth vs oracle", atol=5e-5)


@pytest.mark.parametrize("model,preset", [("triplane", "R1"), ("triplane", "R2"), ("infoinv", "R1")])
def test_full_frame_at_the_references_eval_shape(model, preset):
    """What the reference's evaluation actually launches (VERDICT r4 missing #3): ``renderer(rays, field, chunk=4096, N_samples=-1, ...)``
    (TriPlane/main.py:94) -- N_samples = -1 resolves to the model's own nSamples = 884 (FieldBase.py:71-72,127) -- on a field that carries an
    alpha mask like every trained model (FieldBase.py:261-267), here the one the repo's own ``updateAlphaMask((256,256,256))`` (main.py:330-331)
    builds from the seeded field.  Full 800x800 frame at the module's default level: size-independent properties, the mask's effect, and a
    strided check against the oracle marching the same 884 steps through the same mask.  InfoInv: the same call in InfoInv/main.py (renderer with
    N_samples=-1; its mask from ``updateAlphaMask(reso_mask, infoinv=infoinv)``, InfoInv/main.py:325)."""
    from ngf_amd import synth
    from ngf_amd.fieldbase import renderer
    g, params, step = big_case(model, preset)
    g["gauge_on"] = np.array(1)
    tri = model == "triplane"
    kw = {"iteration": 30001} if tri else {"infoinv": True}
    rays_np = synth.lookat_rays(800, 800)
    rays = torch.from_numpy(rays_np).cuda()
    f = field_for_case(g, params, None, bake=True, bake_color=True) if tri else field_for_case(g, params, None)
    assert f.nSamples == 884
    f.updateAlphaMask((256, 256, 256), **({} if tri else {"infoinv": True}))
    vol = f.alphaMask.alpha_volume
    occ = float(vol.mean())
    assert 0.0 < occ <= 1.0
    rgb, depth = renderer(rays, f, chunk=4096, N_samples=-1, white_bg=True, device="cuda", **({} if tri else kw))
    with torch.no_grad():
        f(rays, N_samples=-1, white_bg=True, collect_stats=True, **kw)
    st = f.last_stats.cpu().numpy()
    assert st[3] == 640000 and st[1] <= st[0] <= 640000 * 884 and st[1] > 0
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all()) and float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    # rays are independent, whatever tile / launch they are in: the reference's own 4096-ray chunks give the same bits
    for c0 in (0, 4096 * 77, 640000 - 4096):
        sub = f(rays[c0:c0 + 4096], N_samples=-1, white_bg=True, **kw)
        assert torch.equal(sub["rgb_map"], rgb[c0:c0 + 4096]) and torch.equal(sub["depth_map"], depth[c0:c0 + 4096])
    idx = torch.arange(0, 640000, 97, device="cuda")
    sub = f(rays[idx], N_samples=-1, white_bg=True, **kw)
    assert torch.equal(sub["rgb_map"], rgb[idx]) and torch.equal(sub["depth_map"], depth[idx])
    # strided oracle check: the same 884 steps, the same packbits mask
    mask = (f.alphaMask.packed_bits(), tuple(int(v) for v in vol.shape[-3:]), f.alphaMask.aabb.cpu().numpy())
    orc = oracle_for_case(g, params, step, mask)
    pick = idx.cpu().numpy()[::12]
    o_rgb, o_depth = orc.render(rays_np[pick], 884)
    _close(rgb[torch.from_numpy(pick).cuda()].cpu().numpy(), o_rgb, "S=884 + mask: rgb vs oracle")
    _close(depth[torch.from_numpy(pick).cuda()].cpu().numpy(), o_depth, "S=884 + mask: depth vs oracle", atol=5e-5)
    # without the mask the samples in cells the mask calls empty are evaluated as well: never fewer, and (max-pooled alpha >= thres keeps every
    # cell that matters) nearly the same picture
    f.alphaMask = None
    f.invalidate()
    nomask = f(rays[idx], N_samples=-1, white_bg=True, collect_stats=True, **kw)
    assert float((nomask["rgb_map"] - rgb[idx]).abs().max()) < 2e-2
    f.release()


def test_handle_rebuilds_reuse_pooled_buffers_and_render_the_same_bits():
    """Round 5: ngf_field_destroy parks a handle's buffers in a pool, the next create of the same shapes takes them from there (a rebuild after a
    parameter change: no hipMalloc / hipFree).  Rebuilds render the same bits, changed parameters are seen, the free memory does not creep, and
    ngf_pool_trim hands the parked buffers back to the driver."""
    from ngf_amd import _lib
    L = _lib.lib()
    g, params, step, mask = load_case("triplane_r1_gauge")
    f = field_for_case(g, params, mask, bake=True, bake_color=True)
    rays = torch.from_numpy(g["rays"]).cuda()
    S = int(g["S"])
    ref = f(rays, N_samples=S, white_bg=True, iteration=30001)
    torch.cuda.synchronize()
    L.ngf_pool_trim()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(12):
        f.invalidate()
        out = f(rays, N_samples=S, white_bg=True, iteration=30001)
        assert torch.equal(out["rgb_map"], ref["rgb_map"]) and torch.equal(out["depth_map"], ref["depth_map"]), k
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 <= 2 * int(L.ngf_field_bytes(f.handle())) + (64 << 20), (free0, free1)          # round 5: one handle in use + one parked, never twelve; round 6: the old image is released first
    with torch.no_grad():
        f.plane_xy.mul_(1.5)                                   # a real change: new pixels from a recycled set of buffers
    out = f(rays, N_samples=S, white_bg=True, iteration=30001)
    assert not torch.equal(out["rgb_map"], ref["rgb_map"])
    with torch.no_grad():
        f.plane_xy.div_(1.5)
    f.invalidate()
    back = f(rays, N_samples=S, white_bg=True, iteration=30001)
    assert float((back["rgb_map"] - ref["rgb_map"]).abs().max()) < 1e-5
    f.release()
    assert L.ngf_pool_trim() == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free1
    f2 = field_for_case(g, params, mask, bake=True, bake_color=True)          # after a trim: plain allocations again
    again = f2(rays, N_samples=S, white_bg=True, iteration=30001)
    assert torch.equal(again["rgb_map"], ref["rgb_map"])
    f2.release()


def test_handle_pool_is_keyed_by_the_handles_device_and_evicts_oldest_first():
    """Round 6: a handle remembers the device it was created on and the streams it was used on.
    ngf_field_destroy waits for those streams (no device-wide synchronisation) and parks the buffers under THE HANDLE'S device id, whatever is
    current in the calling thread; a full pool evicts its oldest entries instead of refusing new ones; a shape change (up_sampling) returns the
    stale sizes to the driver; a destroy issued right behind a render on a side stream waits for that render."""
    from ngf_amd import _lib
    L = _lib.lib()
    g, params, step, mask = load_case("triplane_r1_gauge")
    rays = torch.from_numpy(g["rays"]).cuda()
    S = int(g["S"])
    L.ngf_pool_trim()
    assert L.ngf_pool_bytes(-1) == 0
    f = field_for_case(g, params, mask, bake=True, bake_color=True)
    ref = f(rays, N_samples=S, white_bg=True, iteration=30001)
    hb = int(L.ngf_field_bytes(f.handle()))
    dev = torch.cuda.current_device()
    # destroy while a render is still running on a side stream: the buffers are parked only after that stream's work is done, so the
    # next create (which rewrites them on the current stream) cannot race the render; the pixels of both are the reference's
    side = torch.cuda.Stream()
    big = rays.repeat(64, 1)
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())
        out_side = f(big, N_samples=S, white_bg=True, iteration=30001)
    f.release()
    parked = int(L.ngf_pool_bytes(dev))
    assert 0 < parked <= hb and L.ngf_pool_bytes(dev + 1) == 0 and L.ngf_pool_bytes(-1) == parked      # under the handle's device id
    with torch.no_grad():
        f.plane_xy.mul_(2.0)
    changed = f(rays, N_samples=S, white_bg=True, iteration=30001)          # rebuilt INTO the parked buffers
    assert L.ngf_pool_bytes(dev) < parked
    torch.cuda.synchronize()
    n = rays.shape[0]
    assert torch.equal(out_side["rgb_map"][:n], ref["rgb_map"]) and torch.equal(out_side["rgb_map"][-n:], ref["rgb_map"])
    assert not torch.equal(changed["rgb_map"], ref["rgb_map"])
    # steady state of rebuilds: nothing but the one image is resident (the old handle is released BEFORE the new one is created)
    L.ngf_pool_trim()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        f.invalidate()
        f(rays, N_samples=S, white_bg=True, iteration=30001)
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] <= (48 << 20) and L.ngf_pool_bytes(dev) <= (1 << 20)
    # a byte cap: the oldest parked buffers leave, new ones are still taken
    f.release()
    assert L.ngf_pool_bytes(dev) == parked
    assert L.ngf_pool_set_limit(parked // 2) == 0 and L.ngf_pool_bytes(dev) <= parked // 2
    f2 = field_for_case(g, params, mask, bake=True, bake_color=True)
    again = f2(rays, N_samples=S, white_bg=True, iteration=30001)
    assert torch.equal(again["rgb_map"], ref["rgb_map"])
    f2.release()
    assert 0 < L.ngf_pool_bytes(dev) <= parked // 2
    assert L.ngf_pool_set_limit(0) == 0 and L.ngf_pool_bytes(-1) == 0
    assert L.ngf_pool_set_limit(1 << 30) == 0
    # a shape change trims: after up_sampling the 256^2-sized buffers can never match again
    f3 = field_for_case(g, params, mask, bake=True, bake_color=True)
    f3(rays, N_samples=S, white_bg=True, iteration=30001)
    f3.up_sampling((f3.plane_xy.shape[3] + 8, f3.plane_xy.shape[2] + 8, f3.plane_yz.shape[2] + 8))
    f3(rays, N_samples=S, white_bg=True, iteration=30001)
    assert L.ngf_pool_bytes(-1) == 0
    f3.release()
    L.ngf_pool_trim()


def test_tile_queue_slots_are_clean_after_every_kind_of_launch():
    """a render launch no longer clears its tile-queue slot up front -- the launch's last workgroup zeroes it for the slot's next
    use, 256 launches later.  600 launches that alternate the production instantiation, the debug one (statistics) and three launch shapes
    walk every slot at least twice: a dirty slot would hand out wrong tile numbers and change pixels."""
    g, params, step, mask = load_case("triplane_r1_gauge")
    f = field_for_case(g, params, mask, bake=True, bake_color=True)
    rays = torch.from_numpy(g["rays"]).cuda()
    S = int(g["S"])
    shapes = [rays, rays[:37], rays.repeat(9, 1)]
    refs = [f(r, N_samples=S, white_bg=True, iteration=30001) for r in shapes]
    torch.cuda.synchronize()
    bad = 0
    for k in range(600):
        j = k % 3
        out = f(shapes[j], N_samples=S, white_bg=True, iteration=30001, collect_stats=(k % 5 == 0))
        bad += int(not (torch.equal(out["rgb_map"], refs[j]["rgb_map"]) and torch.equal(out["depth_map"], refs[j]["depth_map"])))
    assert bad == 0
    f.release()


def test_image_tile_order_renders_the_same_bits():
    """Round 6: ngf_field_render_image walks an image-shaped ray list in screen-space blocks (80 rows x 80 pixels: the tiles in
    flight together cover a compact window, so their colour-plane taps stay inside an XCD's L2).  A tile computes what it always did: the pixels of
    the blocked walk are the row-major walk's bit for bit -- full frame, a width the blocks do not divide, a frame with a ragged last row, widths
    the plan cannot use (not a multiple of the 8-ray tile: silently the list's order), the per-XCD queues on top, InfoInv, and the statistics."""
    from ngf_amd import _lib, synth
    from ngf_amd.fieldbase import renderer
    g, params, step = big_case("triplane", "R2")
    f = field_for_case(g, params, None, bake=True, bake_color=True)
    kw = dict(N_samples=96, white_bg=True, iteration=30001)
    for H, W, extra in ((400, 800, 0), (203, 424, 0), (150, 200, 77), (64, 100, 0), (32, 4096, 0)):
        rays = torch.from_numpy(np.concatenate([synth.lookat_rays(H, W), synth.lookat_rays(1, W)[:extra]], 0)).cuda()
        ref = f(rays, collect_stats=True, **kw)
        st0 = f.last_stats.clone()
        out = f(rays, row_width=W, collect_stats=True, **kw)
        assert torch.equal(out["rgb_map"], ref["rgb_map"]) and torch.equal(out["depth_map"], ref["depth_map"]), (H, W)
        assert torch.equal(f.last_stats[:4], st0[:4])
        prod = f(rays, row_width=W, **kw)                     # the production instantiation (no statistics)
        assert torch.equal(prod["rgb_map"], ref["rgb_map"]) and torch.equal(prod["depth_map"], ref["depth_map"]), (H, W)
        with _lib.knobs(xcd=1):
            x = f(rays, row_width=W, **kw)
        assert torch.equal(x["rgb_map"], ref["rgb_map"])
        with _lib.knobs(ord_rows=16, ord_px=32):
            y = f(rays, row_width=W, **kw)
        assert torch.equal(y["rgb_map"], ref["rgb_map"]) and torch.equal(y["depth_map"], ref["depth_map"])
        with _lib.knobs(ord_rows=0):                          # knob: the list's own order
            z = f(rays, row_width=W, **kw)
        assert torch.equal(z["rgb_map"], ref["rgb_map"])
    rays = torch.from_numpy(synth.lookat_rays(120, 160)).cuda()
    rgb, depth = renderer(rays, f, chunk=4096, N_samples=96, white_bg=True, row_width=160)           # the reference's call, plus the hint
    ref = f(rays, **kw)
    assert torch.equal(rgb, ref["rgb_map"]) and torch.equal(depth, ref["depth_map"])
    f.release()
    gi, pi, _ = big_case("infoinv", "R1")
    fi = field_for_case(gi, pi, None)
    rays = torch.from_numpy(synth.lookat_rays(160, 320)).cuda()
    a = fi(rays, N_samples=64, infoinv=True)
    b = fi(rays, N_samples=64, infoinv=True, row_width=320)
    assert torch.equal(a["rgb_map"], b["rgb_map"]) and torch.equal(a["depth_map"], b["depth_map"])
    fi.release()


def test_overlapping_launches_on_two_render_streams_give_the_same_bits():
    """Round 6: bench.py at N > 1 (and ngf_amd.dist.render_streams for any caller) marches consecutive frames on two alternating HIP streams, so the
    launches of ONE field handle overlap -- the next frame's workgroups start on the CUs the previous frame has left.  Every launch takes its own
    tile-queue slot (256 in flight per handle) and writes its own output buffers: twelve frames of three shapes on alternating streams, both
    models, are the serial frames bit for bit, and the queue slots are clean afterwards (the next serial frames are right as well)."""
    from ngf_amd import dist as ndist, synth
    for model, kw in (("triplane", dict(N_samples=96, white_bg=True, iteration=30001)), ("infoinv", dict(N_samples=64, white_bg=True, infoinv=True))):
        g, params, step = big_case(model, "R1")
        f = field_for_case(g, params, None, bake=True, bake_color=True) if model == "triplane" else field_for_case(g, params, None)
        shapes = [torch.from_numpy(synth.lookat_rays(h, w)).cuda() for h, w in ((100, 800), (37, 264), (3, 1000))]
        refs = [f(r, **kw) for r in shapes]
        streams = ndist.render_streams(torch.device("cuda"))
        outs = [(torch.full((shapes[k % 3].shape[0], 3), -1.0, device="cuda"), torch.full((shapes[k % 3].shape[0],), -1.0, device="cuda")) for k in range(12)]
        torch.cuda.synchronize()
        for k in range(12):
            with torch.cuda.stream(streams[k % 2]):
                f(shapes[k % 3], out=outs[k], row_width=(800, 264, 1000)[k % 3], **kw)
        torch.cuda.synchronize()
        for k in range(12):
            assert torch.equal(outs[k][0], refs[k % 3]["rgb_map"]) and torch.equal(outs[k][1], refs[k % 3]["depth_map"]), (model, k)
        again = [f(r, **kw) for r in shapes]
        for a, b in zip(again, refs):
            assert torch.equal(a["rgb_map"], b["rgb_map"]) and torch.equal(a["depth_map"], b["depth_map"])
        f.release()


def test_xcd_tile_queues_are_bit_identical_and_xcds_are_visible():
    """Round 3: with knob xcd = 1 a render launch keeps one tile queue per XCD (each XCD has its own L2; its waves then
    work on ONE compact ray range and steal from the other chunks at the end).  Which wave renders a tile does not change the tile: knob xcd = 0 / 1 give the same bits, on the
    full frame (all 8 queues + stealing), on a launch smaller than the resident waves, and on ragged sizes.  The kernels read the XCD id from
    HW_REG_XCC_ID: ngf_debug_xcd_histogram must see all eight ids, evenly."""
    from ngf_amd import _lib, synth
    L = _lib.lib()
    hist = torch.zeros(8, dtype=torch.int32, device="cuda")
    _lib.check(L.ngf_debug_xcd_histogram(hist.data_ptr(), 256, torch.cuda.current_stream().cuda_stream))
    h = hist.cpu().numpy()
    assert h.sum() == 256 and (h > 0).all() and h.max() <= 2 * h.min(), h
    g, params, step = big_case("triplane", "R1")
    rays = torch.from_numpy(synth.lookat_rays(800, 800)).cuda()
    for bake in (True, False):
        f = field_for_case(g, params, None, bake=bake)
        for n in (640000, 100003, 4096, 37):
            with _lib.knobs(xcd=0):
                a = f(rays[:n], N_samples=192, white_bg=True, iteration=30001)
            with _lib.knobs(xcd=1):
                b = f(rays[:n], N_samples=192, white_bg=True, iteration=30001)
            c = f(rays[:n], N_samples=192, white_bg=True, iteration=30001)          # the launch code's own choice
            assert torch.equal(a["rgb_map"], b["rgb_map"]) and torch.equal(a["depth_map"], b["depth_map"]), (bake, n)
            assert torch.equal(a["rgb_map"], c["rgb_map"]) and torch.equal(a["depth_map"], c["depth_map"]), (bake, n)
        f.release()


def test_infoinv_full_frame_properties():
    """BASELINE config 3 at full size (InfoInv, 800x800, S=192, dense preset, infoinv=True): the size-independent properties of
    test_full_frame_properties + a strided oracle check, for the default kernel AND NGF_F_SPLIT_BF16 (VERDICT r2 missing #4)."""
    from ngf_amd import synth
    g, params, step = big_case("infoinv", "R1")
    g["infoinv"] = np.array(1)
    rays_np = synth.lookat_rays(800, 800)
    rays = torch.from_numpy(rays_np).cuda()
    idx = torch.arange(0, 640000, 97, device="cuda")
    pick = idx.cpu().numpy()[::8]
    orc = oracle_for_case(g, params, step, None)
    o_rgb, o_depth = orc.render(rays_np[pick], 192)
    frames = {}
    for split in (False, True):
        f = field_for_case(g, params, None, split_bf16=split)
        full = f(rays, N_samples=192, white_bg=True, infoinv=True, collect_stats=True)
        st = f.last_stats.cpu().numpy()
        assert st[3] == 640000 and 0.05 < st[1] / (640000 * 192) < 0.5, st
        rgb, depth = full["rgb_map"], full["depth_map"]
        assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
        assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
        sub = f(rays[idx], N_samples=192, white_bg=True, infoinv=True)          # rays are independent: a subset gives the same bits
        assert torch.equal(sub["rgb_map"], rgb[idx]) and torch.equal(sub["depth_map"], depth[idx])
        black = f(rays[idx], N_samples=192, white_bg=False, infoinv=True)["rgb_map"]
        assert bool((sub["rgb_map"] + 1e-6 >= black).all())
        sel = torch.from_numpy(pick).cuda()
        _close(rgb[sel].cpu().numpy(), o_rgb, f"InfoInv full-frame rgb vs oracle (split={split})")
        _close(depth[sel].cpu().numpy(), o_depth, f"InfoInv full-frame depth vs oracle (split={split})", atol=5e-5)
        frames[split] = (rgb, depth)
        f.release()
    # split-bf16 frame against the fp32-MFMA frame: rounding noise on almost every pixel; a sample whose weight sits within an ulp of the
    # 1e-4 colour threshold may be classified differently by the two density evaluations (SURVEY section 7 hazard 3: <= 1e-4 x rgb)
    d = (frames[True][0] - frames[False][0]).abs().max(dim=1).values
    assert float(d.max()) < 1.5e-4 and float((d > 1e-5).float().mean()) < 1e-3, (float(d.max()), float((d > 1e-5).float().mean()))


def test_alpha_mask_build_and_ray_filter():
    """Row N2: getDenseAlpha / updateAlphaMask / filtering_rays against the reference's own outputs."""
    g, params, step, _ = load_case("triplane_alpha_mask")
    f = field_for_case(g, params, None)
    f.alphaMask_thres = float(g["alphaMask_thres"])
    mgrid = tuple(int(v) for v in g["mgrid"])
    alpha, dense_xyz = f.getDenseAlpha(mgrid)
    np.testing.assert_allclose(alpha.cpu().numpy(), g["dense_alpha"], rtol=2e-4, atol=2e-7)
    # the same values through the C oracle (gauge off, identity split of the normalised position)
    orc = oracle_for_case(g, params, step, None)
    xyz = dense_xyz.view(-1, 3).cpu().numpy()
    xn = (xyz - g["aabb"][0]) * (np.float32(2.0) / (g["aabb"][1] - g["aabb"][0])) - np.float32(1.0)
    coords = np.stack([xn[:, 0], xn[:, 1], xn[:, 1], xn[:, 2], xn[:, 0], xn[:, 2]], 1).astype(np.float32)
    o_alpha = 1.0 - np.exp(-orc.density_at(coords) * np.float32(step))
    np.testing.assert_allclose(alpha.view(-1).cpu().numpy(), o_alpha, rtol=2e-4, atol=2e-7)
    new_aabb = f.updateAlphaMask(mgrid)
    # the one-call build evaluates the same lattice: its dense alpha is getDenseAlpha's, transposed (FieldBase.py:185)
    assert torch.equal(f.last_dense_alpha, alpha.transpose(0, 2).contiguous())
    assert np.array_equal(f.alphaMask.packed_bits_device().cpu().numpy(), f.alphaMask.packed_bits())      # device bit-pack = np.packbits
    vol = f.alphaMask.alpha_volume[0, 0].cpu().numpy()
    near_thr = np.abs(torch.nn.functional.max_pool3d(torch.from_numpy(g["dense_alpha"]).clamp(0, 1).transpose(0, 2)[None, None], 3, 1, 1)[0, 0].numpy()
                      - float(g["alphaMask_thres"])) < 1e-6
    assert np.array_equal(vol[~near_thr], g["mask_volume"][~near_thr])
    np.testing.assert_allclose(new_aabb.cpu().numpy(), g["new_aabb"], atol=1e-6)
    rays, rgbs = torch.from_numpy(g["rays"]), torch.from_numpy(g["rgbs"])
    kept, kept_rgb = f.filtering_rays(rays, rgbs, N_samples=40)
    assert kept.shape == g["kept_rays"].shape and np.array_equal(kept.numpy(), g["kept_rays"])
    kept_b, _ = f.filtering_rays(rays, rgbs, bbox_only=True)
    assert np.array_equal(kept_b.numpy(), g["kept_bbox"])
    # the rebuilt mask is used by the next render and by a checkpoint round trip
    out = f(rays.cuda(), N_samples=32, iteration=30001)
    assert bool(torch.isfinite(out["rgb_map"]).all())


@pytest.mark.parametrize("dhw,keep", [((5, 6, 7), None), ((9, 4, 13), 0.6), ((3, 3, 3), 2.0), ((4, 5, 2), -1.0), ((16, 16, 16), 0.9)])
def test_alpha_mask_sign_at_arbitrary_points(dhw, keep):
    """Row A4 on its own: `sample_alpha(xyz) > 0` (FieldBase.py:33-40, 263-267 -- the sign of ATen's 3-D grid_sample on the {0,1} volume) at arbitrary
    points through compute_alpha (alpha = 0 exactly where the mask is empty), against the reference's own grid_sample outputs
    (tests/golden/ops_grid_sample.npz: a blobby 5 x 6 x 7 volume, points in and around it) and against the C oracle on blobby, full and empty
    volumes of other shapes: random points, points far outside, and the lattice points themselves (weights exactly 0 / 1: boundary cells decided
    by the weighted sum).  The kernels read ONE byte per sample -- the 8 corner bits of the sample's cell (mask_cells_kernel) -- so none / all / mixed
    cells and the one-cell border with zero-padded corners are all in here."""
    import ctypes as C
    from oracle import oracle as O
    from ngf_amd import synth
    fx = np.load(os.path.join(GOLDEN, "ops_grid_sample.npz"))
    if keep is None:
        bits, q_fix, want_fix = np.ascontiguousarray(fx["mask_bits"]), fx["q"], fx["out3"] > 0
        assert tuple(int(v) for v in fx["mask_dhw"]) == dhw
    else:
        _, bits = synth.alpha_mask_bits(17, dhw, keep=keep)          # keep = 2: every voxel set (the coarse pattern still carves holes); keep = -1: empty
        if keep >= 2.0:
            bits = np.packbits(np.ones(int(np.prod(dhw)), np.uint8))
        q_fix, want_fix = None, None
    d, h, w = dhw
    # not the field's box, not symmetric -- except for the fixture's volume, whose points are normalised coordinates: the box [-1, 1]^3 hands most of them through unchanged
    maabb = np.array([[-1.0, -0.8, -1.1], [1.0, 0.9, 1.3]] if keep is not None else [[-1.0] * 3, [1.0] * 3], np.float32)
    g0, params, _ = big_case("triplane", "R1", res=32)
    f = field_for_case(g0, params, (bits, dhw, maabb))
    rng = np.random.default_rng(5)
    pts = [rng.uniform(-1.45, 1.45, (20000, 3)).astype(np.float32)]
    lat = np.stack(np.meshgrid(np.linspace(maabb[0, 0], maabb[1, 0], w, dtype=np.float32), np.linspace(maabb[0, 1], maabb[1, 1], h, dtype=np.float32),
                               np.linspace(maabb[0, 2], maabb[1, 2], d, dtype=np.float32), indexing="ij"), -1).reshape(-1, 3)
    pts += [lat, lat + np.float32(1e-6), lat - np.float32(1e-6)]
    pts.append(np.array([[1e6, 0, 0], [-1e6, 0, 0], [0, 3e38, 0], [0, 0, -3e38], [1.0, 0.9, 1.3], [-1.0, -0.8, -1.1]], np.float32))
    if q_fix is not None:
        pts.append(q_fix.astype(np.float32))
    p = np.ascontiguousarray(np.concatenate(pts, 0).astype(np.float32))
    alpha = f.compute_alpha(torch.from_numpy(p), length=1000.0).cpu().numpy()
    got = alpha != 0
    # the reference's normalisation (AlphaGridMask.normalize_coord, FieldBase.py:39-40) in IEEE fp32, then the oracle's grid_sample
    inv = (np.float32(1.0) / (maabb[1] - maabb[0]) * np.float32(2.0)).astype(np.float32)
    qn = np.ascontiguousarray(((p - maabb[0]) * inv - np.float32(1.0)).astype(np.float32))
    out3 = np.zeros((p.shape[0],), np.float32)
    O.lib().ngf_oracle_mask_sample(bits.ctypes.data_as(C.c_void_p), d, h, w, qn.ctypes.data_as(C.c_void_p), C.c_int64(p.shape[0]), out3.ctypes.data_as(C.c_void_p))
    want = out3 > 0
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {p.shape[0]} points differ"
    if keep is not None and keep < 0:
        assert not got.any()
    else:
        assert got.any() and not got.all()
    if q_fix is not None:
        # where the normalisation (p + 1) * 1 - 1 reproduces the fixture's point bit for bit, the REFERENCE's own sign must come out
        back = qn[-q_fix.shape[0]:]
        same = np.all(back == q_fix, axis=1)
        assert same.sum() > 50
        assert np.array_equal(got[-q_fix.shape[0]:][same], want_fix[same])


def _sparse_masks(name):
    """Occupancy volumes with real empty space, [D,H,W] on the field's +-1.5 box: a ball (an object in empty space), two small slabs near opposite
    corners (long empty runs between them), and a blobby random volume (little to skip: the test of the conservative side)."""
    if name == "ball":
        ax = np.linspace(-1.5, 1.5, 128, dtype=np.float32)
        zz, yy, xx = np.meshgrid(ax, ax, ax, indexing="ij")
        return (xx ** 2 + yy ** 2 + zz ** 2) < 0.8 ** 2
    if name == "slabs":
        vol = np.zeros((200, 96, 256), bool)          # anisotropic: cells per step differ per axis
        vol[10:24, 8:30, 20:60] = True
        vol[150:190, 60:90, 180:250] = True
        vol[100, 48, 128] = True                       # a single voxel in the middle of nowhere
        return vol
    if name == "lattice":                              # thin walls every 24 cells inside a ball: empty cells everywhere, hardly an 8-cell block that is clear -- the 4-cell image's case
        ax = np.linspace(-1.5, 1.5, 160, dtype=np.float32)
        zz, yy, xx = np.meshgrid(ax, ax, ax, indexing="ij")
        wall = (np.arange(160) % 24) < 1
        return (wall[:, None, None] | wall[None, :, None] | wall[None, None, :]) & ((xx ** 2 + yy ** 2 + zz ** 2) < 1.2 ** 2)
    from ngf_amd import synth
    return synth.alpha_mask_bits(23, (64, 64, 64), keep=0.35)[0]


@pytest.mark.parametrize("model,level", [("triplane", 3), ("triplane", 2), ("triplane", "bf16"), ("infoinv", "split")])
@pytest.mark.parametrize("mask_name", ["ball", "slabs", "blobby", "lattice"])
def test_empty_space_skipping_is_bit_identical(model, level, mask_name):
    """Round 6: after an iteration without a valid sample the march asks the mask's block images (8^3 and, when those certify nothing, 4^3 cells per block; a block is `clear` when it and its 26
    neighbours hold no occupied corner) how many of the next steps sample empty cells for certain, and jumps over whole iterations of them
    (ngf_render.hpp, mask_clear_around).  Skipped samples have sigma = alpha = w = 0 in the reference too (FieldBase.py:261-270), so nothing may change:
    the frame with the skip against the frame without it (DBG knob ablate = 128) bit for bit, for every tile shape (rays per wave tile 64 ... 1: the run
    length is read from the ballot differently per shape), with per-ray jitter (train-mode forward), the same number of evaluated samples, the
    production instantiation the same bits again, and a strided subset against the oracle at the reference's S = 884."""
    from ngf_amd import synth, triplane
    from ngf_amd._lib import knobs
    g, params, step = big_case(model, "R1")
    kw_f = {3: dict(bake=True, bake_color=True), 2: dict(bake=True), "bf16": dict(bake=True, bake_color=True, split_bf16=True), "split": dict(split_bf16=True)}[level]
    f = field_for_case(g, params, None, **kw_f)
    vol = _sparse_masks(mask_name)
    f.alphaMask = triplane.AlphaGridMask("cuda", torch.tensor(np.asarray(g["aabb"], np.float32)), torch.from_numpy(vol.astype(np.float32)).cuda())
    f.invalidate()
    full = synth.lookat_rays(800, 800).reshape(800, 800, 6)
    rays_np = np.ascontiguousarray(full[3::11, 5::11].reshape(-1, 6))          # 73 x 73 rays over the whole frame (many miss the object, many graze it)
    rays = torch.from_numpy(rays_np).cuda()
    kw = {"iteration": 30001} if model == "triplane" else {"infoinv": True}
    S = -1                                                                      # the model's own nSamples: 884
    jit = torch.from_numpy(synth.hash_uniform(31, 7, (rays_np.shape[0],)).astype(np.float32))
    ref = None
    for tw in ((8, 64, 32, 16, 4, 2, 1) if model == "triplane" else (8, 16, 4, 2, 1)):          # (the InfoInv bf16 kernel takes tiles of at most 16 rays)
        with knobs(ablate=128, tile_w=tw):
            off = f(rays, N_samples=S, white_bg=True, collect_stats=True, **kw)
        n_off = f.last_stats.cpu().numpy().copy()
        with knobs(tile_w=tw):
            on = f(rays, N_samples=S, white_bg=True, collect_stats=True, **kw)
            prod = f(rays, N_samples=S, white_bg=True, **kw)
        n_on = f.last_stats.cpu().numpy().copy()
        for a, b in ((off, on), (off, prod)):
            assert torch.equal(a["rgb_map"], b["rgb_map"]) and torch.equal(a["depth_map"], b["depth_map"]), f"tile_w {tw}"
        assert n_on[0] == n_off[0] and n_on[1] == n_off[1], f"tile_w {tw}: evaluated / active samples {n_on[:2]} vs {n_off[:2]}"
        if ref is None:
            ref = off
        if tw <= 8:          # the tile shapes of a production launch give the same bits (wider tiles fold the view inputs differently: other last bits of the colours)
            assert torch.equal(ref["rgb_map"], on["rgb_map"]) and torch.equal(ref["depth_map"], on["depth_map"]), f"tile_w {tw} vs 8"
        with torch.no_grad():
            with knobs(ablate=128, tile_w=tw):
                t_off = f(rays, N_samples=S, white_bg=True, is_train=True, jitter=jit, coin=0.9, collect_stats=True, **kw)
            with knobs(tile_w=tw):
                t_on = f(rays, N_samples=S, white_bg=True, is_train=True, jitter=jit, coin=0.9, **kw)
        assert torch.equal(t_off["rgb_map"], t_on["rgb_map"]) and torch.equal(t_off["depth_map"], t_on["depth_map"]), f"tile_w {tw}, jitter"
    assert 0 < n_on[0] < 0.5 * rays_np.shape[0] * 884          # there IS empty space in these volumes
    # rays that are not a camera's: origins inside and outside the box, directions of any length (0.2 ... 3: the cells-per-step bound uses |d|), some with
    # zero components, some pointing away from the box -- with and without the skip, same bits
    rng = np.random.default_rng(3)
    ro = rng.uniform(-2.5, 2.5, (6000, 3)).astype(np.float32)
    rd = rng.normal(size=(6000, 3)).astype(np.float32)
    rd *= (rng.uniform(0.2, 3.0, (6000, 1)) / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    rd[::7, 0] = 0.0
    rd[::11, 1:] = 0.0
    odd = torch.from_numpy(np.concatenate([ro, rd], 1)).cuda()
    for tw in (8, 1) if model == "triplane" else (16, 1):
        with knobs(ablate=128, tile_w=tw):
            a = f(odd, N_samples=S, white_bg=False, collect_stats=True, **kw)
        with knobs(tile_w=tw):
            b = f(odd, N_samples=S, white_bg=False, **kw)
        assert torch.equal(a["rgb_map"], b["rgb_map"]) and torch.equal(a["depth_map"], b["depth_map"]), f"odd rays, tile_w {tw}"
    # ... and the oracle on a subset (its mask test is the restated grid_sample)
    sub = slice(0, rays_np.shape[0], 9)
    bits = np.packbits(vol.reshape(-1))
    orc = oracle_for_case(dict(g, model=np.array(model)), params, step, (bits, vol.shape, np.asarray(g["aabb"], np.float32)))
    o_rgb, o_depth = orc.render(rays_np[sub], 884, white_bg=True)
    tol = dict(rtol=1e-4, atol=1e-5) if level in (2, 3) else dict(rtol=2e-4, atol=4e-5)
    np.testing.assert_allclose(ref["rgb_map"].cpu().numpy()[sub], o_rgb, **tol)
    np.testing.assert_allclose(ref["depth_map"].cpu().numpy()[sub], o_depth, rtol=1e-4, atol=5e-5)


def test_split_bf16_fields_build_the_same_alpha_mask():
    """Density queries of NGF_F_SPLIT_BF16 fields (compute_alpha / getDenseAlpha): TriPlane's density path is untouched by the flag
    (bit-identical); InfoInv's density MLP runs on split bf16 products too -- the reference's own dense alpha at the usual tolerance."""
    for name in ("infoinv_alpha_mask", "triplane_alpha_mask"):
        g, params, step, _ = load_case(name)
        mgrid = tuple(int(v) for v in g["mgrid"])
        fa, fb = field_for_case(g, params, None), field_for_case(g, params, None, split_bf16=True)
        a, _ = fa.getDenseAlpha(mgrid)
        b, _ = fb.getDenseAlpha(mgrid)
        if name.startswith("triplane"):
            assert torch.equal(a, b), name
        else:
            np.testing.assert_allclose(b.cpu().numpy(), g["dense_alpha_on"], rtol=2e-4, atol=2e-7)
            np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-5, atol=1e-7)
            pts = torch.from_numpy(g["pts"]).cuda()
            np.testing.assert_allclose(fb.compute_alpha(pts, 0.37).cpu().numpy(), g["alpha_pts_on"], rtol=2e-4, atol=2e-7)


def test_infoinv_alpha_api_takes_the_infoinv_flag():
    """InfoInv tree: compute_alpha / getDenseAlpha / updateAlphaMask(..., infoinv=True|False) (InfoInv/models/FieldBase.py:140-193;
    InfoInv/main.py:325 calls field.updateAlphaMask(tuple(reso_mask), infoinv=infoinv)) against the reference's own outputs in
    BOTH modes -- with infoinv=False the occupancy mask must come from the un-modulated densities."""
    g, params, step, _ = load_case("infoinv_alpha_mask")
    mgrid = tuple(int(v) for v in g["mgrid"])
    pts = torch.from_numpy(g["pts"]).cuda()
    for flag, tag in ((True, "on"), (False, "off")):
        f = field_for_case(g, params, None)
        f.alphaMask_thres = float(g["alphaMask_thres"])
        a = f.compute_alpha(pts, 0.37, infoinv=flag)
        np.testing.assert_allclose(a.cpu().numpy(), g["alpha_pts_" + tag], rtol=2e-4, atol=2e-7)
        alpha, _ = f.getDenseAlpha(mgrid, infoinv=flag)
        np.testing.assert_allclose(alpha.cpu().numpy(), g["dense_alpha_" + tag], rtol=2e-4, atol=2e-7)
        f.updateAlphaMask(mgrid, infoinv=flag)
        vol = f.alphaMask.alpha_volume[0, 0].cpu().numpy()
        pooled = torch.nn.functional.max_pool3d(torch.from_numpy(g["dense_alpha_" + tag]).clamp(0, 1).transpose(0, 2)[None, None], 3, 1, 1)[0, 0].numpy()
        near_thr = np.abs(pooled - float(g["alphaMask_thres"])) < 1e-6
        assert np.array_equal(vol[~near_thr], g["mask_volume_" + tag][~near_thr])
        assert 0.02 < vol.mean() < 0.98
    assert np.abs(g["dense_alpha_on"] - g["dense_alpha_off"]).max() > 1e-2            # the two modes really differ
    # default = infoinv=True, like the reference's signatures
    f = field_for_case(g, params, None)
    assert torch.equal(f.compute_alpha(pts, 0.37), f.compute_alpha(pts, 0.37, infoinv=True))
    # the TriPlane tree has no such keyword (TriPlane/models/FieldBase.py:140)
    gt, pt, _, _ = load_case("triplane_r1_gauge")
    with pytest.raises(TypeError):
        field_for_case(gt, pt, None).compute_alpha(pts, 1.0, infoinv=True)


def test_renders_a_checkpoint_written_by_the_reference():
    """The reference's own checkpoint file, loaded as is, renders the reference's own pixels."""
    import os
    from ngf_amd import triplane
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = torch.load(os.path.join(ROOT, "tests", "golden", "ref_ckpt_triplane.th"), map_location="cpu", weights_only=False)
    kw = dict(ck["kwargs"])
    kw.update({"device": "cuda"})
    f = triplane.TriPlane(**kw)
    f.load(ck)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_ckpt_triplane.npz"))
    out = f(torch.from_numpy(g["rays"]).cuda(), white_bg=True, is_train=False, N_samples=40, iteration=30001)
    np.testing.assert_allclose(out["rgb_map"].cpu().numpy(), g["rgb_map"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["depth_map"].cpu().numpy(), g["depth_map"], rtol=1e-4, atol=1e-5)


def test_steady_state_calls_do_not_synchronise_with_the_host():
    """The reference's drivers keep ``aabb`` on the GPU (TriPlane/main.py:211).  Round 4 found ``aabb.tolist()`` in the per-call handle key: a
    device-to-host copy + stream synchronisation in front of every render call and training step of such a caller.  torch's sync debug mode
    turns every synchronising torch call into an error: the second render of a field, and training steps after the first, must pass under it
    (ctypes calls into the library are not seen by that mode; the library's own entry points do not synchronise on these paths by
    construction -- include/ngf.h)."""
    from ngf_amd import train, triplane
    from helpers import big_case
    from ngf_amd import synth
    g, params, step = big_case("triplane", "R1")
    aabb = torch.tensor(np.asarray(g["aabb"], np.float32)).cuda()                 # on the device, like scene_bbox.to(device)
    f = triplane.TriPlane(aabb, [int(v) for v in g["grid"]], "cuda", gauge_start=0, near_far=[2.0, 6.0], alphaMask_thres=1e-4, distance_scale=25.0,
                          rayMarch_weight_thres=1e-4, step_ratio=0.5)
    f.load_params(params)
    frame = torch.from_numpy(synth.lookat_rays(800, 800)[:8192]).cuda()
    tgt = torch.from_numpy(synth.hash_uniform(5, 2, (4096, 3))).cuda()
    jit = torch.from_numpy(synth.hash_uniform(5, 3, (4096,))).cuda()
    ref = f(frame, N_samples=64, white_bg=True, iteration=30001)["rgb_map"].clone()
    tr = train.Trainer(f, batch_size=4096, max_samples=64)
    tr.step(frame[:4096], tgt, 0, N_samples=64, jitter=jit)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = f(frame, N_samples=64, white_bg=True, iteration=30001)["rgb_map"]      # parameters changed by the step: the handle is rebuilt here
        out2 = f(frame, N_samples=64, white_bg=True, iteration=30001)["rgb_map"]
        tr.step(frame[:4096], tgt, 1, N_samples=64, jitter=jit)
        tr.step(frame[4096:], tgt, 2, N_samples=64, jitter=jit)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and not torch.equal(out, ref)
