"""CPU: the eval-output-stage oracle (oracle/evalout.py) against the vectors captured from the reference's own
rgb_ssim / visualize_depth_numpy / evaluation arithmetic (tests/golden/evalout.npz, made by make_golden.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import evalout as orc  # noqa: E402
import ngf_amd  # noqa: E402,F401
from ngf_amd import evalout  # noqa: E402

G = dict(np.load(os.path.join(ROOT, "tests", "golden", "evalout.npz")))


def test_ssim_matches_reference():
    assert abs(orc.rgb_ssim(G["img0"], G["img1"], 1) - float(G["ssim"])) < 1e-13
    np.testing.assert_allclose(orc.rgb_ssim(G["img0"], G["img1"], 1, return_map=True), G["ssim_map"], rtol=0, atol=1e-13)
    assert abs(orc.rgb_ssim(G["img0"], G["img1"], 1, filter_size=5, filter_sigma=0.8) - float(G["ssim5"])) < 1e-13
    assert abs(orc.rgb_ssim(G["img0"], G["img0"], 1) - 1.0) < 1e-12


def test_mse_psnr_match_reference():
    assert abs(orc.mse(G["img0"], G["img1"]) - float(G["mse"])) < 1e-7 * float(G["mse"])       # torch's float32 mean vs float64 sum
    assert abs(orc.psnr(G["img0"], G["img1"]) - float(G["psnr"])) < 1e-5


def test_depth_index_and_u8_match_reference():
    idx, _ = orc.depth_index(G["depth"], (2.0, 6.0))
    assert np.array_equal(idx, G["depth_idx_nearfar"])
    idx, rng = orc.depth_index(G["depth_finite"], None)
    assert np.array_equal(idx, G["depth_idx_auto"])
    np.testing.assert_array_equal(np.asarray(rng, np.float64), G["depth_auto_range"])
    assert np.array_equal(orc.frame_u8(G["rgb"]), G["rgb8"])


def test_jet_table_shape_and_landmarks():
    """The table itself is parity-unpinned (no cv2 in the image); these are the published landmarks of COLORMAP_JET."""
    lut = evalout.jet_lut()
    assert lut.shape == (256, 3) and lut.dtype == np.uint8
    assert tuple(lut[0]) == (128, 0, 0) and tuple(lut[255]) == (0, 0, 128)         # B,G,R: dark blue -> dark red
    assert tuple(lut[95]) == (255, 252, 0) and tuple(lut[96]) == (254, 255, 2)      # red starts rising at index 96
    assert lut[:, 1].max() == 255 and lut[128, 1] == 255
    # every ramp moves in steps of 4 levels
    d = np.abs(np.diff(lut.astype(int), axis=0))
    assert set(np.unique(d)) <= {0, 1, 2, 3, 4}


def test_jet_table_reproduces_the_published_entries():
    """evalout.jet_lut against the entries OpenCV prints for Jet::r (colormap.cpp): the closed form is the table, and the uint8
    conversion (cvRound(255 v), half to even) of those entries is what the LUT holds; symmetry gives the other channels."""
    from ngf_amd import evalout
    lut = evalout.jet_lut()
    assert lut.shape == (256, 3) and lut.dtype == np.uint8
    for k, v in enumerate(evalout.PUBLISHED_JET_R_96):
        i = 96 + k
        assert abs((1.5 - abs(4.0 * i / 255.0 - 3.0)) - v) < 1e-15
        # OpenCV holds the entries as float32 and converts with cvRound(float32(v) * 255.0f) (convertTo CV_32F -> CV_8U): every ramp
        # entry lands on an exact .5 in float32 and rounds half to even -- the halves arithmetic of jet_lut
        p = np.float32(v) * np.float32(255.0)
        assert p - np.floor(p) == 0.5 and lut[i, 2] == int(np.rint(p))
    assert np.all(lut[:96, 2] == 0) and lut[255, 2] == 128 and lut[0, 0] == 128          # dark blue -> dark red, 0.5 at both ends
    assert np.array_equal(lut[:, 0], lut[::-1, 2])                                        # blue is red mirrored
    assert np.array_equal(lut[:, 1], lut[::-1, 1]) and lut[127, 1] == 255 and lut[128, 1] == 255
    # the whole table through OpenCV's float32 pipeline (entries as float32, x 255.0f in float32, cvRound)
    i = np.arange(256, dtype=np.float64)
    for col, c in ((0, 1.0), (1, 2.0), (2, 3.0)):
        v32 = np.clip(1.5 - np.abs(4.0 * i / 255.0 - c), 0.0, 1.0).astype(np.float32)
        assert np.array_equal(np.rint(v32 * np.float32(255.0)).astype(np.uint8), lut[:, col])
