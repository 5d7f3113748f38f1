"""Ray-sharded multi-GPU render (new; the reference is single-device apart from one nn.DataParallel
wrapper, UV-Mapping/model/model.py:285).

Rays of a frame are independent (no cross-ray term anywhere in Base.forward), so rank r renders the
contiguous block [r*ceil(N/W), ...) with replicated parameters and the only exchange is ONE all-gather
of the composited pixels ([n_rank, 4] fp32: rgb + depth; 10 MB per 800x800 frame) -- RCCL over xGMI
when the process group's backend is "nccl".  One process per GPU, launched by torch.distributed.run.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous, equal-size shards (the last ranks may own padding): [lo, hi) and the common shard size."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


def render_sharded(render_fn, rays, group=None, gather: bool = True):
    """``render_fn(rays_shard) -> (rgb [m,3], depth [m])`` on this rank's device.

    Returns (rgb [N,3], depth [N]) on every rank when ``gather`` (all_gather_into_tensor of one packed
    [per,4] buffer per rank), else this rank's shard only.  ``rays`` is the full [N,6] tensor (host or
    device); only the local slice is moved / rendered.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = rays.shape[0]
    lo, hi, per = shard_bounds(n, world, rank)
    rgb, depth = render_fn(rays[lo:hi])
    if not gather or world == 1:
        return rgb, depth
    return gather_pixels(rgb, depth, n, per, world, group)


def shard_buffers(per: int, device):
    """One [4*per] float32 send buffer whose first 3*per floats are the rgb view and last per the depth
    view, so a field can render straight into the all-gather operand (no packing copy)."""
    buf = torch.zeros((4 * per,), device=device, dtype=torch.float32)
    return buf, buf[: 3 * per].view(per, 3), buf[3 * per:]


def gather_pixels(rgb, depth, n: int, per: int, world: int, group=None, send=None, recv=None):
    """All-gather of the composited pixels.  ``send`` (from shard_buffers) avoids the packing copy."""
    if send is None:
        send, r_view, d_view = shard_buffers(per, rgb.device)
        r_view[: rgb.shape[0]] = rgb
        d_view[: depth.shape[0]] = depth
    if recv is None:
        recv = torch.empty((world * 4 * per,), device=send.device, dtype=torch.float32)
    dist.all_gather_into_tensor(recv, send, group=group)
    blocks = recv.view(world, 4 * per)
    full_rgb = blocks[:, : 3 * per].reshape(world * per, 3)[:n]
    full_depth = blocks[:, 3 * per:].reshape(world * per)[:n]
    return full_rgb, full_depth
