"""GPU: seeded sweep of small random configurations of the render path against the C oracle -- plane shapes (rectangular,
tiny), sample counts, ray counts that straddle the tile widths, gauge on/off, background, alpha mask, density presets from
fog to opaque walls, both models.  Guards the kernel's launch-shape logic (split tiles, early termination, empty-iteration
skip, per-ray view fold) beyond the golden cases."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import field_for_case, oracle_for_case  # noqa: E402
import ngf_amd  # noqa: E402,F401
from ngf_amd import geometry, synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _case(k):
    u = synth.hash_uniform(1000 + k, 1, (16,))
    model = "infoinv" if k % 4 == 3 else "triplane"
    grid = [int(3 + u[0] * 30), int(3 + u[1] * 30), int(3 + u[2] * 30)]
    plane_hw = ((grid[1], grid[0]), (grid[2], grid[1]), (grid[2], grid[0]))
    gauge_hw = (int(2 + u[3] * 14), int(2 + u[4] * 14))
    preset = ("R0", "R1", "R2")[int(u[5] * 3) % 3]
    g = {"model": np.array(model), "aabb": np.array([[-1.5, -1.4, -1.6], [1.5, 1.6, 1.3]], np.float32), "grid": np.array(grid),
         "near_far": np.array([2.0, 6.0], np.float32), "step_ratio": np.float32(0.5 + u[6]), "distance_scale": np.float32(25),
         "thr": np.float32(1e-4), "gauge_on": np.array(int(u[7] < 0.7)), "infoinv": np.array(int(u[8] < 0.7))}
    if model == "triplane":
        params = synth.triplane_params(2000 + k, plane_hw, gauge_hw, preset=preset, gauge_std=float(0.002 + 0.1 * u[9]))
        if u[10] < 0.3:                                            # an opaque wall: early termination after a few steps
            params["density_decoder.bias"] = np.array([22.0], np.float32)
    else:
        params = synth.infoinv_params(2000 + k, plane_hw, preset=preset)
    mask = None
    if u[11] < 0.4:
        dhw = (int(3 + u[12] * 10), int(3 + u[13] * 10), int(3 + u[14] * 10))
        _, bits = synth.alpha_mask_bits(3000 + k, dhw)
        mask = (bits, dhw, np.array([[-1.4, -1.3, -1.5], [1.4, 1.5, 1.2]], np.float32))
    n = [1, 3, 7, 8, 9, 31, 33, 64, 65, 130, 257, 300][k % 12]
    frame = synth.lookat_rays(24, 24)
    pick = (synth.hash_uniform(4000 + k, 1, (n,)) * np.float32(frame.shape[0])).astype(np.int64)
    rays = frame[pick]
    m = rays[::3].shape[0]
    edge = synth.edge_rays(5000 + k, max(4, 4 * ((m + 3) // 4)))
    rays[::3] = edge[:m]
    S = [1, 2, 7, 15, 16, 17, 40, 63, 64, 65, 97, 128][(k * 5) % 12]
    return g, params, mask, rays, S, bool(u[15] < 0.6)


@pytest.mark.parametrize("variant", ["level1", "split_bf16", "level3"])
@pytest.mark.parametrize("k", range(24))
def test_random_configuration_matches_oracle(k, variant):
    """split_bf16: the same sweep through NGF_F_SPLIT_BF16 (colour MLP on the bf16 matrix pipe with 3-term split operands,
    csrc/ngf_shade_bf16.hpp / ngf_infoinv.hpp); level3: through the module's default since round 4 (density_decoder and layer 1 folded into
    the planes, quad-coalesced gather, 12-float queue records) -- same oracle, same tolerances, same threshold-adjacency rule."""
    g, params, mask, rays, S, white = _case(k)
    step = geometry.step_size(g["aabb"], [int(v) for v in g["grid"]], float(g["step_ratio"]))
    orc = oracle_for_case(g, params, step, mask)
    if variant == "level3" and str(g["model"]) != "triplane":
        pytest.skip("levels are a TriPlane option")
    f = field_for_case(g, params, mask, split_bf16=variant == "split_bf16", bake=variant == "level3", bake_color=variant == "level3")
    kw = {"iteration": 30001 if int(g["gauge_on"]) else -1} if str(g["model"]) == "triplane" else {"infoinv": bool(int(g["infoinv"]))}
    if str(g["model"]) == "triplane":
        f.gauge_start = 0
    out = f(torch.from_numpy(rays).cuda(), N_samples=S, white_bg=white, **kw)
    o_rgb, o_depth = orc.render(rays, S, white_bg=white)
    rgb, depth = out["rgb_map"].cpu().numpy(), out["depth_map"].cpu().numpy()
    bad_rgb = np.abs(rgb - o_rgb) > (1e-5 + 1e-4 * np.abs(o_rgb))
    bad_depth = np.abs(depth - o_depth) > (1e-5 + 1e-4 * np.abs(o_depth))
    # depth_map has no threshold in it (sum of ALL weights x z, FieldBase.py:305-306) and the sample positions / masks are
    # bit-exact, so nothing may be forgiven there
    assert not bad_depth.any(), (k, int(bad_depth.sum()), float(np.abs(depth - o_depth).max()))
    # rgb_map: the only legitimate source of an out-of-tolerance pixel is a sample whose weight sits at the colour threshold
    # (weight > 1e-4, FieldBase.py:289) and is classified differently by the two expf implementations.  Every such pixel must
    # be explained by the oracle's own per-sample weights of that ray: n_near samples within rounding distance of the threshold
    # (the transmittance is a product of up to S factors of 1 ulp each), and a difference of at most n_near flipped samples.
    thr = float(g["thr"])
    rel = (S + 8) * 2.0 ** -23
    for r in np.unique(np.nonzero(bad_rgb)[0]):
        _, _, dbg = orc.render(rays[r:r + 1], S, white_bg=white, debug_rays=1)
        w = dbg["weight"][0].astype(np.float64)
        n_near = int(np.sum(np.abs(w - thr) <= rel * thr))
        worst = float(np.abs(rgb[r] - o_rgb[r]).max())
        assert n_near >= 1, f"case {k} ray {r}: |rgb - oracle| = {worst:.3e} with no weight within {rel:.1e} (relative) of the threshold"
        assert worst <= n_near * 1.05 * thr + 1e-5, f"case {k} ray {r}: {worst:.3e} is more than {n_near} threshold flips can move a pixel"
    # determinism and batch independence
    again = f(torch.from_numpy(rays).cuda(), N_samples=S, white_bg=white, **kw)
    assert torch.equal(out["rgb_map"], again["rgb_map"]) and torch.equal(out["depth_map"], again["depth_map"])
    if rays.shape[0] > 4:
        part = f(torch.from_numpy(rays[2:-1]).cuda(), N_samples=S, white_bg=white, **kw)
        assert torch.equal(out["rgb_map"][2:-1], part["rgb_map"]) and torch.equal(out["depth_map"][2:-1], part["depth_map"])
