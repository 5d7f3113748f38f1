"""On-device ray generation: the DTU loader's get_rays_dir (UV-Mapping/data/dtu.py:27-37) and, for a pin-hole camera, get_ray_directions + get_rays
(TriPlane/dataLoader/ray_utils.py:24-42, 66-87) with the Blender loader's normalisation (blender.py:52) -- so a
frame (or one rank's row block of it) never crosses PCIe as a [H*W,6] tensor."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib


def blender_focal(W: int, camera_angle_x: float = 0.6911112070083618) -> float:
    """blender.py:46-47: focal = 0.5*800/tan(0.5*camera_angle_x) * (W/800) in double, rounded to float32 once (where torch
    divides the float32 pixel grid by it, ray_utils.py:40)."""
    import numpy as np
    return float(np.float32(0.5 * 800 / math.tan(0.5 * camera_angle_x) * (W / 800)))


def generate_rays(H: int, W: int, focal: float, c2w, rows=None, device="cuda") -> torch.Tensor:
    """rays [rows*W, 6] on ``device`` for image rows [r0, r1); c2w = 3x4 (OpenCV axes) array-like."""
    r0, r1 = (0, H) if rows is None else rows
    c = torch.as_tensor(c2w, dtype=torch.float32).reshape(-1)[:12].tolist()
    out = torch.empty(((r1 - r0) * W, 6), device=device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().ngf_generate_rays(H, W, C.c_float(focal), (C.c_float * 12)(*c), r0, r1 - r0, out.data_ptr(),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def generate_rays_dtu(H: int, W: int, focal, princpt, rot, rows=None, device="cuda") -> torch.Tensor:
    """raydir [rows*W, 3] on ``device``: get_rays_dir on the full-image pixel grid (UV-Mapping/data/dtu.py:27-37,160-168);
    focal [2], princpt [2], rot [3,3] = extrinsics[view][:3,:3] as the dataset's in_cam*.npy hold them."""
    r0, r1 = (0, H) if rows is None else rows
    f = torch.as_tensor(focal, dtype=torch.float32).reshape(-1)[:2].tolist()
    c = torch.as_tensor(princpt, dtype=torch.float32).reshape(-1)[:2].tolist()
    r = torch.as_tensor(rot, dtype=torch.float32).reshape(-1)[:9].tolist()
    out = torch.empty(((r1 - r0) * W, 3), device=device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().ngf_generate_rays_dtu(H, W, (C.c_float * 2)(*f), (C.c_float * 2)(*c), (C.c_float * 9)(*r), r0, r1 - r0,
                                                    out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out
