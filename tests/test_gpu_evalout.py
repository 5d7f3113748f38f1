"""GPU: the eval output stage (csrc/ngf_eval.hpp through the C ABI, host mirror ngf_amd/evalout.py) against the
oracle and the vectors captured from the reference (tests/golden/evalout.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import evalout as orc  # noqa: E402
import ngf_amd  # noqa: E402,F401
from ngf_amd import evalout, synth  # noqa: E402

pytestmark = pytest.mark.gpu
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "evalout.npz")))


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_frame_u8_bit_exact():
    assert np.array_equal(evalout.to_uint8(cu(G["rgb"])).cpu().numpy(), G["rgb8"])
    edge = np.array([0.0, 1.0, 1 / 255, 0.99999994, 254.999 / 255, -0.0, 2.0, -3.0, 0.5, np.nan], np.float32)
    assert np.array_equal(evalout.to_uint8(cu(edge)).cpu().numpy(), orc.frame_u8(edge))


def test_depth_colormap_bit_exact():
    lut = evalout.jet_lut()
    img, rng = evalout.visualize_depth_numpy(cu(G["depth"]), (2.0, 6.0))
    assert rng == [2.0, 6.0]
    assert np.array_equal(img.cpu().numpy(), lut[G["depth_idx_nearfar"]])           # reference index image + the table
    img, rng = evalout.visualize_depth_numpy(cu(G["depth_finite"]), None)
    np.testing.assert_array_equal(rng.cpu().numpy().astype(np.float64), G["depth_auto_range"])
    assert np.array_equal(img.cpu().numpy(), lut[G["depth_idx_auto"]])
    # identity table = the index image itself, at a ragged size
    ident = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)
    d = (synth.hash_uniform(5, 1, (1037,)) * np.float32(9) - np.float32(1)).astype(np.float32)
    img, _ = evalout.visualize_depth_numpy(cu(d), (2.0, 6.0), cmap=ident)
    assert np.array_equal(img.cpu().numpy()[:, 0], orc.depth_index(d, (2.0, 6.0))[0])


def test_mse_psnr():
    m = evalout.mse(cu(G["img0"]), cu(G["img1"])).item()
    assert abs(m - float(G["mse"])) < 1e-7 * float(G["mse"])
    assert abs(m - orc.mse(G["img0"], G["img1"])) < 1e-12 * m
    assert abs(evalout.psnr(cu(G["img0"]), cu(G["img1"])) - float(G["psnr"])) < 1e-5


def test_ssim_matches_reference_and_oracle():
    a, b = cu(G["img0"]), cu(G["img1"])
    assert abs(evalout.rgb_ssim(a, b, 1) - float(G["ssim"])) < 1e-12
    np.testing.assert_allclose(evalout.rgb_ssim(a, b, 1, return_map=True).cpu().numpy(), G["ssim_map"], rtol=0, atol=1e-12)
    assert abs(evalout.rgb_ssim(a, b, 1, filter_size=5, filter_sigma=0.8) - float(G["ssim5"])) < 1e-12
    assert abs(evalout.rgb_ssim(a, a, 1) - 1.0) < 1e-12
    with pytest.raises(RuntimeError):
        evalout.rgb_ssim(a[:8], b[:8], 1)                  # image smaller than the filter
    with pytest.raises(RuntimeError):
        evalout.rgb_ssim(torch.from_numpy(G["img0"]), torch.from_numpy(G["img1"]), 1)       # host tensors: no CPU path


def test_full_frame_sizes():
    """800x800: deterministic, SSIM(x,x)=1, MSE symmetric, oracle agreement on a strip."""
    H = W = 800
    x = synth.hash_uniform(6, 1, (H, W, 3))
    y = np.clip(x + (synth.hash_uniform(6, 2, (H, W, 3)) - np.float32(0.5)) * np.float32(0.1), 0, 1).astype(np.float32)
    a, b = cu(x), cu(y)
    s1, s2 = evalout.rgb_ssim(a, b, 1), evalout.rgb_ssim(a, b, 1)
    assert s1 == s2 and 0.0 < s1 < 1.0
    assert abs(evalout.rgb_ssim(a, a, 1) - 1.0) < 1e-12
    assert evalout.mse(a, b).item() == evalout.mse(b, a).item()
    full = evalout.rgb_ssim(a, b, 1, return_map=True)
    strip = orc.rgb_ssim(x[100:140], y[100:140], 1, return_map=True)          # rows 100..129 of the map
    np.testing.assert_allclose(full[100:130].cpu().numpy(), strip, rtol=0, atol=1e-12)
    assert abs(float(full.mean().item()) - s1) < 1e-12
    assert np.array_equal(evalout.to_uint8(a).cpu().numpy(), orc.frame_u8(x))


def test_frame_outputs_on_a_render():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case, field_for_case
    g, params, step, mask = load_case("triplane_r1_gauge")
    f = field_for_case(g, params, mask)
    rays = torch.from_numpy(g["rays"][:192]).cuda()
    r = f(rays, N_samples=48, iteration=30001)
    gt = (r["rgb_map"] * 0.9 + 0.05).clamp(0, 1)
    o = evalout.frame_outputs(r["rgb_map"], r["depth_map"], 12, 16, (2.0, 6.0), gt)
    assert o["rgb8"].shape == (12, 16, 3) and o["depth8"].shape == (12, 16, 3) and o["rgbd8"].shape == (12, 32, 3)
    rgb = r["rgb_map"].cpu().numpy().reshape(12, 16, 3)
    assert np.array_equal(o["rgb8"].cpu().numpy(), orc.frame_u8(rgb))
    assert abs(o["psnr"] - orc.psnr(np.clip(rgb, 0, 1), gt.cpu().numpy().reshape(12, 16, 3))) < 1e-6
    assert abs(o["ssim"] - orc.rgb_ssim(np.clip(rgb, 0, 1), gt.cpu().numpy().reshape(12, 16, 3), 1)) < 1e-12
    idx, _ = orc.depth_index(r["depth_map"].cpu().numpy().reshape(12, 16), (2.0, 6.0))
    assert np.array_equal(o["depth8"].cpu().numpy(), evalout.jet_lut()[idx])


def test_evaluation_path_renders_poses(tmp_path):
    """evaluation_path (main.py:141-183): device ray generation per pose + renderer + output stage; frame 0 equals the
    frame rendered from host-built rays of the same pose."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case, field_for_case
    from ngf_amd import rays as nrays
    g, params, step, mask = load_case("triplane_r1_gauge")
    f = field_for_case(g, params, mask)
    H = W = 24
    ds = types.SimpleNamespace(near_far=[2.0, 6.0], img_wh=(W, H), focal=nrays.blender_focal(W))
    poses = [synth.lookat_pose(azim_deg=a) for a in (40.0, 100.0)]
    frames, depths = evalout.evaluation_path(ds, f, poses, savePath=str(tmp_path), N_samples=48, white_bg=True)
    assert len(frames) == 2 and frames[0].shape == (H, W, 3) and frames[0].dtype == torch.uint8 and depths[1].shape == (H, W, 3)
    assert (tmp_path / "000.png").exists() and (tmp_path / "rgbd" / "001.png").exists()
    host = torch.from_numpy(synth.lookat_rays(H, W, c2w=poses[0])).cuda()
    ref = f(host, N_samples=48, white_bg=True, iteration=30001)["rgb_map"]
    diff = (evalout.to_uint8(ref).reshape(H, W, 3).int() - frames[0].int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.01          # device-built rays differ in the last ulp
    assert not torch.equal(frames[0], frames[1])


def test_evaluation_mirror_on_a_tiny_dataset(tmp_path):
    """evalout.evaluation = TriPlane/main.py:73-138 without LPIPS / MP4: per test view renderer + device output stage, PSNR list,
    PNGs and mean.txt.  Ground truth = the field's own render (+ a constant offset), so the PSNR is known in closed form."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case, field_for_case
    g, params, step, mask = load_case("triplane_r1_gauge")
    f = field_for_case(g, params, mask)
    H = W = 20
    views = [torch.from_numpy(synth.lookat_rays(H, W, c2w=synth.lookat_pose(azim_deg=a))) for a in (20.0, 80.0, 200.0, 290.0)]
    with torch.no_grad():
        gts = [(f(v.cuda(), N_samples=40, white_bg=True, iteration=30001)["rgb_map"].clamp(0, 1) * 0.5 + 0.25).cpu() for v in views]
    ds = types.SimpleNamespace(near_far=[2.0, 6.0], img_wh=(W, H), all_rays=torch.stack(views), all_rgbs=torch.stack(gts))
    psnrs = evalout.evaluation(ds, f, None, savePath=str(tmp_path), N_vis=2, prtx="t_", N_samples=40, white_bg=True)
    assert len(psnrs) == 2 and all(np.isfinite(psnrs))                     # N_vis=2 of 4 views -> views 0 and 2
    for k, idx in enumerate((0, 2)):
        rgb = f(views[idx].cuda(), N_samples=40, white_bg=True, iteration=30001)["rgb_map"].clamp(0, 1).cpu()
        want = -10.0 * np.log(torch.mean((rgb - gts[idx]) ** 2).item()) / np.log(10.0)
        assert abs(psnrs[k] - want) < 1e-4
    assert (tmp_path / "t_000.png").exists() and (tmp_path / "rgbd" / "t_001.png").exists()
    vals = np.loadtxt(tmp_path / "t_mean.txt")
    assert vals.shape == (2,) and abs(vals[0] - np.mean(psnrs)) < 1e-6 and 0.0 < vals[1] <= 1.0
