"""Pins the CPU restatement (oracle/ngf_oracle.c) against outputs of the reference itself.

The reference has no tests or golden vectors of its own; the fixtures in
tests/golden/ were captured by importing /root/reference with torch 2.10.0 CPU
(tests/golden/make_golden.py).  Tolerance: 1e-6 abs on pixels --
the only differences are summation order and libm-vs-SLEEF last-ulp effects.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import load_case, oracle_for_case
from oracle import oracle as O

TRIPLANE = ["triplane_r1_gauge", "triplane_r2_nogauge", "triplane_r1_mask", "triplane_r0"]
INFOINV = ["infoinv_r1_on", "infoinv_r1_off", "infoinv_r1_mask"]           # _mask: InfoInv WITH an alpha mask, black background (round 4)
TRAIN = ["triplane_r1_train_white", "triplane_r1_train_black"]             # is_train=True forwards: supplied jitter, background coin (round 4)


def case_render_args(g):
    """(white_bg, jitter) of a captured forward: a training-mode forward (FieldBase.py:128-130, 299) carries its per-ray jitter and the
    background coin -- white_bg or coin < 0.5."""
    white = bool(int(g["white_bg"]))
    if "is_train" in g:
        return white or float(g["coin"]) < 0.5, g["jitter"]
    return white, None


@pytest.mark.parametrize("name", TRIPLANE + INFOINV + TRAIN)
def test_render_matches_reference(name):
    g, params, step, mask = load_case(name)
    assert np.float32(step) == np.float32(g["stepSize"]), "init_para stepSize restatement differs"
    orc = oracle_for_case(g, params, step, mask)
    white, jitter = case_render_args(g)
    rgb, depth, dbg = orc.render(g["rays"], int(g["S"]), white_bg=white, jitter=jitter, debug_rays=8)
    assert np.max(np.abs(rgb - g["rgb_map"])) <= 2e-6
    assert np.max(np.abs(depth - g["depth_map"])) <= 5e-6
    # intermediates of the first 8 rays: positions and masks are bit-exact, the rest to rounding
    assert np.array_equal(dbg["z"], g["i_z"])
    assert np.array_equal(dbg["valid"].astype(bool), g["i_valid"])
    np.testing.assert_allclose(dbg["sigma"], g["i_sigma"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(dbg["alpha"], g["i_alpha"], rtol=1e-4, atol=2e-7)
    np.testing.assert_allclose(dbg["weight"], g["i_weight"], rtol=1e-4, atol=2e-7)
    if "i_coords" in g:
        np.testing.assert_allclose(dbg["coords"], g["i_coords"], rtol=0, atol=5e-7)
    # per-sample colours (FieldBase.py:289-294): the reference's rgb_mask and its own compute_rgb / rgb_decoder outputs on the first 8 rays
    m = g["i_rgb_mask"]
    near = np.abs(g["i_weight"] - np.float32(g["thr"])) < 1e-7
    assert np.array_equal((dbg["weight"] > np.float32(g["thr"]))[~near], m[~near])
    if m.any():
        dirs = np.broadcast_to(g["rays"][:8, None, 3:6], (*m.shape, 3))[m]
        col = orc.color_at(np.ascontiguousarray(g["i_coords"][m]), np.ascontiguousarray(dirs))
        assert np.max(np.abs(col - g["i_rgb"][m])) <= 2e-6
    assert not g["i_rgb"][~m].any()


def test_nsamples_restatement():
    from ngf_amd import geometry
    for name in TRIPLANE[:1] + INFOINV[:1]:
        g, _, step, _ = load_case(name)
        assert geometry.n_samples(g["aabb"], step) == int(g["nSamples"])
    # headline geometry: 256^3 grid, +-1.5 box, step_ratio 0.5
    aabb = [[-1.5] * 3, [1.5] * 3]
    s = geometry.step_size(aabb, [256] * 3, 0.5)
    assert abs(float(s) - 0.0058823532) < 1e-9 and geometry.n_samples(aabb, s) == 884


def test_samplers_match_aten(golden_dir):
    g = np.load(golden_dir + "/ops_grid_sample.npz")
    lib = O.lib()
    plane = np.ascontiguousarray(g["plane"][0])
    uv = np.ascontiguousarray(g["uv"])
    out = np.zeros((uv.shape[0], plane.shape[0]), np.float32)
    lib.ngf_oracle_bilerp2d(plane.ctypes.data_as(C.c_void_p), plane.shape[1], plane.shape[2], plane.shape[0],
                            uv.ctypes.data_as(C.c_void_p), C.c_int64(uv.shape[0]), out.ctypes.data_as(C.c_void_p))
    assert np.max(np.abs(out - g["out2"])) <= 5e-7
    q = np.ascontiguousarray(g["q"])
    bits = np.ascontiguousarray(g["mask_bits"])
    d, h, w = (int(v) for v in g["mask_dhw"])
    out3 = np.zeros((q.shape[0],), np.float32)
    lib.ngf_oracle_mask_sample(bits.ctypes.data_as(C.c_void_p), d, h, w, q.ctypes.data_as(C.c_void_p),
                               C.c_int64(q.shape[0]), out3.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out3 > 0, g["out3"] > 0)       # the path only uses the sign (FieldBase.py:264)
    assert np.max(np.abs(out3 - g["out3"])) <= 5e-7


def test_infoinv_compute_alpha_both_modes():
    """compute_alpha(xyz, length, infoinv=True|False) of the InfoInv tree (InfoInv/models/FieldBase.py:140-156) through the C
    restatement's density, against the reference's own outputs (tests/golden/infoinv_alpha_mask.npz)."""
    from helpers import load_case, oracle_for_case
    g, params, step, _ = load_case("infoinv_alpha_mask")
    pts = g["pts"]
    inside = np.all((pts >= g["aabb"][0]) & (pts <= g["aabb"][1]), axis=1)
    xn = (pts - g["aabb"][0]) * (np.float32(2.0) / (g["aabb"][1] - g["aabb"][0])) - np.float32(1.0)
    coords = np.stack([xn[:, 0], xn[:, 1], xn[:, 1], xn[:, 2], xn[:, 0], xn[:, 2]], 1).astype(np.float32)
    for flag, tag in ((1, "on"), (0, "off")):
        g["infoinv"] = np.array(flag)
        orc = oracle_for_case(g, params, step, None)
        alpha = 1.0 - np.exp(-orc.density_at(coords) * np.float32(0.37))
        # outside the box the reference still evaluates the planes (grid_sample zero padding): compare everywhere
        np.testing.assert_allclose(alpha, g["alpha_pts_" + tag], rtol=2e-5, atol=2e-7)
    assert inside.any() and (~inside).any()
