"""The torch-eager CPU port that bench.py times as cpu_baseline must itself match the reference:
checked against the golden vectors captured from the reference and against the C oracle."""
import numpy as np
import pytest
import torch

from helpers import load_case, oracle_for_case
from oracle.eager import EagerField


@pytest.mark.parametrize("name", ["triplane_r1_gauge", "triplane_r2_nogauge", "triplane_r1_mask", "infoinv_r1_on",
                                  "infoinv_r1_off"])
def test_eager_port_matches_reference_golden(name):
    g, params, step, mask = load_case(name)
    am = None
    if mask is not None:
        bits, dhw, maabb = mask
        am = (np.unpackbits(bits)[:int(np.prod(dhw))].reshape(dhw).astype(np.float32), maabb)
    e = EagerField(params, g["aabb"], step, g["near_far"], float(g["distance_scale"]), float(g["thr"]), str(g["model"]), am)
    mode = bool(int(g["gauge_on"])) if "gauge_on" in g else bool(int(g["infoinv"]))
    rgb, depth = e.render(torch.from_numpy(g["rays"]), int(g["S"]), chunk=100, white_bg=bool(int(g["white_bg"])), mode=mode)
    assert np.abs(rgb.numpy() - g["rgb_map"]).max() <= 2e-6
    assert np.abs(depth.numpy() - g["depth_map"]).max() <= 5e-6
    o_rgb, _ = oracle_for_case(g, params, step, mask).render(g["rays"], int(g["S"]), white_bg=bool(int(g["white_bg"])))
    assert np.abs(rgb.numpy() - o_rgb).max() <= 3e-6
