"""Eval output stage on the device -- host-side mirror of the reference's helpers:

    visualize_depth_numpy(depth, minmax=None, cmap=COLORMAP_JET)   TriPlane/utils.py:32-47
    rgb_ssim(img0, img1, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False)  utils.py:109-155
    the per-frame body of evaluation(...)                           TriPlane/main.py:93-121  -> frame_outputs / evaluation

Same names and argument meaning; inputs are device tensors (the renderer's outputs) and the arithmetic runs in
libngf_hip.so (csrc/ngf_eval.hpp) through the C ABI.  There is no CPU fallback.  LPIPS (utils.py:85-97) needs
pretrained networks and stays out; the MP4 writers of the reference need imageio and stay on the caller.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import _lib

COLORMAP_JET = 2          # cv2.COLORMAP_JET


# the first rising entries of Jet::r as printed in OpenCV's colormap.cpp (index 96 .. 104 of 256)
PUBLISHED_JET_R_96 = (0.00588235294117645, 0.02156862745098032, 0.03725490196078418, 0.05294117647058827, 0.06862745098039214,
                      0.084313725490196, 0.1000000000000001, 0.115686274509804, 0.1313725490196078)


def jet_lut() -> np.ndarray:
    """cv2.COLORMAP_JET as a [256,3] uint8 table in B,G,R order.

    OpenCV's `class Jet` (imgproc/src/colormap.cpp; "equals the GNU Octave colormap jet") holds three 256-entry float arrays over the
    breakpoints linspace(0, 1, 256) -- unlike most of its maps, which hold 64 entries that `linear_colormap` interpolates -- so
    `interp1` onto the same 256 points returns the entries themselves, and `convertTo(CV_8U, 255)` rounds them with cvRound (half to
    even).  The printed entries are clamp(1.5 - |4 i/255 - c|, 0, 1) with c = 3 (red), 2 (green), 1 (blue) evaluated in double
    (r[96..104] = 0.00588235294117645, 0.02156862745098032, 0.03725490196078418, 0.05294117647058827, 0.06862745098039214,
    0.084313725490196, 0.1000000000000001, 0.115686274509804, 0.1313725490196078: PUBLISHED_JET_R_96 below, checked in
    tests/test_evalout_oracle.py).  255 v = 382.5 - |4 i - 255 c| is a half-integer on every ramp entry, and so is OpenCV's float32
    product float32(v) * 255.0f (checked for all 256 x 3 entries): both round half to even, so the table is evaluated in halves here.
    Pinned to those published table values, not to a run of cv2.applyColorMap (cv2 is not in the image).
    """
    i = np.arange(256, dtype=np.float64)
    ch = [np.clip(382.5 - np.abs(4.0 * i - 255.0 * c), 0.0, 255.0) for c in (1, 2, 3)]       # B, G, R
    return np.stack([np.rint(v) for v in ch], -1).astype(np.uint8)


_LUTS = {}
_WS = {}


def _lut(device, cmap):
    if cmap != COLORMAP_JET:
        raise NotImplementedError("only cv2.COLORMAP_JET (2) is built in; pass a [256,3] uint8 B,G,R table as `cmap` otherwise")
    key = (str(device), cmap)
    if key not in _LUTS:
        _LUTS[key] = torch.from_numpy(jet_lut()).to(device).contiguous()
    return _LUTS[key]


def _workspace(device, H=0, W=0, fs=0):
    need = int(_lib.lib().ngf_eval_workspace_bytes(H, W, fs))
    key = str(device)
    if key not in _WS or _WS[key].numel() < need:
        _WS[key] = torch.empty((need,), dtype=torch.uint8, device=device)
    return _WS[key]


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor (the eval output stage has no CPU path)")
    return t.contiguous().float()


def to_uint8(rgb_map: torch.Tensor) -> torch.Tensor:
    """``(rgb_map.clamp(0,1).numpy() * 255).astype('uint8')`` (main.py:98,117), same shape, on the device."""
    x = _f32(rgb_map, "rgb_map")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().ngf_eval_frame_u8(x.data_ptr(), x.numel(), out.data_ptr(), _stream(x)))
    return out


def visualize_depth_numpy(depth: torch.Tensor, minmax=None, cmap=COLORMAP_JET):
    """utils.py:32-47.  depth [H,W] device float32 -> ([H,W,3] uint8 device tensor in B,G,R order, [mi, ma]).

    With ``minmax=None`` the range is reduced on the device and returned as a 2-element device tensor (reading it is
    the only thing that would synchronise); with ``minmax=(mi, ma)`` the same pair is returned."""
    x = _f32(depth, "depth")
    dev = x.device
    if isinstance(cmap, (torch.Tensor, np.ndarray)):
        lut = torch.as_tensor(cmap, dtype=torch.uint8).to(dev).contiguous()
        if tuple(lut.shape) != (256, 3):
            raise ValueError("a colour table must be [256,3] uint8 (B,G,R)")
    else:
        lut = _lut(dev, cmap)
    L = _lib.lib()
    if minmax is None:
        rng = torch.empty((2,), dtype=torch.float32, device=dev)
        _lib.check(L.ngf_eval_depth_range(x.data_ptr(), x.numel(), rng.data_ptr(), _workspace(dev).data_ptr(), _stream(x)))
        ret = rng
    else:
        rng = torch.tensor([float(minmax[0]), float(minmax[1])], dtype=torch.float32).to(dev, non_blocking=True)
        ret = [minmax[0], minmax[1]]
    out = torch.empty((*x.shape, 3), dtype=torch.uint8, device=dev)
    _lib.check(L.ngf_eval_depth_colormap(x.data_ptr(), x.numel(), rng.data_ptr(), lut.data_ptr(), out.data_ptr(), _stream(x)))
    return out, ret


def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``torch.mean((a - b) ** 2)`` (main.py:105) -> 0-dim float64 device tensor."""
    a, b = _f32(a, "a"), _f32(b, "b")
    if a.shape != b.shape:
        raise RuntimeError(f"shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    out = torch.empty((1,), dtype=torch.float64, device=a.device)
    _lib.check(_lib.lib().ngf_eval_mse(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _workspace(a.device).data_ptr(), _stream(a)))
    return out[0]


def psnr(rgb_map: torch.Tensor, gt_rgb: torch.Tensor) -> float:
    """``-10 * np.log(loss.item()) / np.log(10)`` (main.py:106)."""
    return float(-10.0 * np.log(mse(rgb_map, gt_rgb).item()) / np.log(10.0))


def rgb_ssim(img0, img1, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False):
    """utils.py:109-155 on the device.  Returns a Python float (mean SSIM) or, with ``return_map``, the
    [H-fs+1, W-fs+1, 3] float64 device tensor."""
    assert len(img0.shape) == 3
    assert img0.shape[-1] == 3
    assert img0.shape == img1.shape
    a, b = _f32(img0, "img0"), _f32(img1, "img1")
    H, W = int(a.shape[0]), int(a.shape[1])
    dev = a.device
    mean = torch.empty((1,), dtype=torch.float64, device=dev)
    smap = torch.empty((H - filter_size + 1, W - filter_size + 1, 3), dtype=torch.float64, device=dev) if return_map else None
    _lib.check(_lib.lib().ngf_eval_ssim(a.data_ptr(), b.data_ptr(), H, W, float(max_val), int(filter_size), float(filter_sigma), float(k1),
                                        float(k2), mean.data_ptr(), smap.data_ptr() if return_map else None,
                                        _workspace(dev, H, W, filter_size).data_ptr(), _stream(a)))
    return smap if return_map else float(mean.item())


def frame_outputs(rgb_map, depth_map, H, W, near_far, gt_rgb=None, compute_extra_metrics=True):
    """The per-frame body of ``evaluation`` (main.py:98-121) on the device: returns a dict with ``rgb8`` [H,W,3] uint8,
    ``depth8`` [H,W,3] uint8 (B,G,R), ``rgbd8`` (the side-by-side image of main.py:123) and, when ``gt_rgb`` is given,
    ``psnr`` (+ ``ssim`` if compute_extra_metrics).  Only the metric scalars are read back."""
    rgb = rgb_map.reshape(H, W, 3)
    depth = depth_map.reshape(H, W)
    out = {}
    depth8, _ = visualize_depth_numpy(depth, near_far)
    if gt_rgb is not None:
        gt = gt_rgb.reshape(H, W, 3).to(rgb.device)
        clamped = rgb.clamp(0.0, 1.0)
        out["psnr"] = psnr(clamped, gt)
        if compute_extra_metrics:
            out["ssim"] = rgb_ssim(clamped, gt, 1)
    out["rgb8"] = to_uint8(rgb)
    out["depth8"] = depth8
    out["rgbd8"] = torch.cat((out["rgb8"], depth8), dim=1)
    return out


@torch.no_grad()
def evaluation(test_dataset, field, args=None, savePath=None, N_vis=5, prtx='', N_samples=-1, white_bg=False,
               compute_extra_metrics=True, device='cuda'):
    """``evaluation`` (TriPlane/main.py:73-138): renders every N_vis-th test view with ``renderer`` (chunk 4096) and
    runs the output stage on the device.  Returns the PSNR list like the reference; PNGs are written with PIL when
    ``savePath`` is given, ``{prtx}mean.txt`` holds [psnr, ssim] (LPIPS is out of scope; no MP4)."""
    from .fieldbase import renderer
    PSNRs, ssims = [], []
    if savePath is not None:
        os.makedirs(savePath, exist_ok=True)
        os.makedirs(savePath + "/rgbd", exist_ok=True)
    near_far = test_dataset.near_far
    n_img = test_dataset.all_rays.shape[0]
    interval = 1 if N_vis < 0 else max(n_img // N_vis, 1)
    idxs = list(range(0, n_img, interval))
    for idx, samples in enumerate(test_dataset.all_rays[0::interval]):
        W, H = test_dataset.img_wh
        rays = samples.view(-1, samples.shape[-1])
        rgb_map, depth_map = renderer(rays, field, chunk=4096, N_samples=N_samples, white_bg=white_bg, device=device, row_width=W)
        gt = test_dataset.all_rgbs[idxs[idx]].view(H, W, 3) if len(test_dataset.all_rgbs) else None
        o = frame_outputs(rgb_map, depth_map, H, W, near_far, gt, compute_extra_metrics)
        if gt is not None:
            PSNRs.append(o["psnr"])
            if compute_extra_metrics:
                ssims.append(o["ssim"])
        if savePath is not None:
            from PIL import Image
            Image.fromarray(o["rgb8"].cpu().numpy()).save(f'{savePath}/{prtx}{idx:03d}.png')
            Image.fromarray(o["rgbd8"].cpu().numpy()).save(f'{savePath}/rgbd/{prtx}{idx:03d}.png')
    if PSNRs and savePath is not None:
        vals = [np.mean(np.asarray(PSNRs))] + ([np.mean(np.asarray(ssims))] if compute_extra_metrics else [])
        np.savetxt(f'{savePath}/{prtx}mean.txt', np.asarray(vals))
    return PSNRs


@torch.no_grad()
def evaluation_path(test_dataset, field, c2ws, savePath=None, N_vis=5, prtx='', N_samples=-1, white_bg=False,
                    compute_extra_metrics=True, device='cuda', focal=None):
    """``evaluation_path`` (TriPlane/main.py:141-183): one frame per camera pose of a path.  The rays of every pose are
    built ON THE DEVICE from (c2w, intrinsics) (ngf_generate_rays) instead of get_rays on the host, rendered
    in one launch and post-processed by the device output stage; returns the list of [H,W,3] uint8 frames (device tensors)
    and the list of depth visualisations.  ``focal`` defaults to ``test_dataset.focal`` (blender.py:47).  PNGs are written
    with PIL when ``savePath`` is given; the MP4 writers of the reference need imageio and stay with the caller."""
    from .fieldbase import renderer
    from .rays import generate_rays
    if savePath is not None:
        os.makedirs(savePath, exist_ok=True)
        os.makedirs(savePath + "/rgbd", exist_ok=True)
    near_far = test_dataset.near_far
    W, H = test_dataset.img_wh
    f = float(focal if focal is not None else test_dataset.focal)
    rgb_maps, depth_maps = [], []
    for idx, c2w in enumerate(c2ws):
        rays = generate_rays(H, W, f, torch.as_tensor(c2w, dtype=torch.float32)[:3, :4], device=device)
        rgb_map, depth_map = renderer(rays, field, chunk=8192, N_samples=N_samples, white_bg=white_bg, device=device, row_width=W)
        o = frame_outputs(rgb_map, depth_map, H, W, near_far)
        rgb_maps.append(o["rgb8"])
        depth_maps.append(o["depth8"])
        if savePath is not None:
            from PIL import Image
            Image.fromarray(o["rgb8"].cpu().numpy()).save(f'{savePath}/{prtx}{idx:03d}.png')
            Image.fromarray(o["rgbd8"].cpu().numpy()).save(f'{savePath}/rgbd/{prtx}{idx:03d}.png')
    return rgb_maps, depth_maps
