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


def test_baseline_config0_plumbing():
    """BASELINE.json configs[0]: TriPlane 64x64 render, 64 samples/ray, CPU eager -- port vs C oracle."""
    from helpers import big_case
    from ngf_amd import synth
    g, params, step = big_case("triplane", "R1", res=64)
    rays = synth.lookat_rays(64, 64)
    e = EagerField(params, g["aabb"], step, g["near_far"], 25.0, 1e-4, "triplane")
    rgb, depth = e.render(torch.from_numpy(rays), 64, chunk=4096)
    g["gauge_on"] = np.array(1)
    o_rgb, o_depth = oracle_for_case(g, params, step, None).render(rays, 64)
    assert np.abs(rgb.numpy() - o_rgb).max() < 3e-6 and np.abs(depth.numpy() - o_depth).max() < 1e-5
    assert rgb.shape == (4096, 3)
