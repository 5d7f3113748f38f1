"""Ray-sharded multi-GPU render (new; the reference is single-device apart from one nn.DataParallel
wrapper, UV-Mapping/model/model.py:285).

Rays of a frame are independent (no cross-ray term anywhere in Base.forward): every rank renders its share with
replicated parameters and the only exchange is ONE all-gather of the composited pixels ([n_rank, 4] fp32: rgb + depth;
10 MB per 800x800 frame) -- RCCL over xGMI when the process group's backend is "nccl".  One process per GPU, launched by
torch.distributed.run.  Two partitions:
  * `render_sharded` / `shard_bounds`: contiguous ray blocks [r*ceil(N/W), ...) -- the generic entry point for any ray list;
  * `interleaved_rows` + `PipelinedGather` (`frame_in_image_order`; `frame` + `deinterleave` is the two-step form) -- what bench.py runs at N > 1: the frame's rows dealt out in 10-row
    blocks round robin.  BASELINE.json's north_star words the partition as contiguous row blocks; contiguous 100-row shards of the
    800x800 frame differ by 7 % in render time (background rows are cheap), and strong scaling pays for the slowest rank, so the
    rows are interleaved and one strided device copy per output (rgb, depth), on a stream of its own, puts the gathered frame back in image order.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous, equal-size shards (the last ranks may own padding): [lo, hi) and the common shard size."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


def interleaved_rows(H: int, world: int, rank: int, block: int):
    """Row ranges [(r0, r1), ...] of rank ``rank`` when the frame is dealt out in blocks of ``block`` rows, round
    robin: rank r owns blocks r, r+world, ...  Every rank then sees the same mix of cheap (background, box edge)
    and expensive rows -- contiguous 100-row shards of the 800x800 frame differ by 7 % in render time
    (profiles/r01_split_march.txt), and the slowest rank is what strong scaling pays for."""
    if H % (world * block):
        raise ValueError(f"H={H} must be a multiple of world*block={world * block}")
    return [(b * block, (b + 1) * block) for b in range(rank, H // block, world)]


def deinterleave(rgb, depth, H: int, W: int, world: int, block: int):
    """Rank-major gathered pixels ([world*per, 3], [world*per], per = H*W/world, each rank's rows in
    interleaved_rows order) -> frame order.  One strided copy on the device."""
    nblk = H // (world * block)
    rgb = rgb.view(world, nblk, block * W, 3).permute(1, 0, 2, 3).reshape(H * W, 3)
    depth = depth.view(world, nblk, block * W).permute(1, 0, 2).reshape(H * W)
    return rgb, depth


# Exchange buffers bench.py keeps in flight at N > 1.  A render launch is a persistent grid that fills every CU (three waves per SIMD at 168 registers),
# so RCCL's all-gather kernel of frame k gets its workgroups placed only where a render workgroup has left -- while the NEXT frame's grid (the other
# render stream) is waiting for the same CUs: the exchange of frame k may complete during the march of frame k + 2 or k + 3.  ``buffers(k)`` makes the
# march of frame k wait for the reorder of frame k - depth.  Measured with a stand-in exchange kernel that holds CUs (profiles/r06_exchange_contention.txt):
# three buffers -- tuned beside a free local-copy exchange -- lose 8-24 % once the exchange holds CUs for 150-400 us, four or six do not; on eight hardware
# queues (GPU_MAX_HW_QUEUES=8, see ``render_streams``) six are never worse than four and 2.5 % better beside a short exchange.  11.5 MB per buffer pair.
PIPELINE_DEPTH = 6


def render_streams(device, n: int = 2):
    """``n`` HIP streams to march consecutive frames on in turn (frame k on stream k % n), each ordered behind the caller's current stream.
    A render launch is a persistent grid of one workgroup per CU; on ONE stream frame k + 1 starts when the last wave of frame k has ended, so
    every frame pays the launch's tail (wave slots idle while the last tiles finish) and the launch gap.  On two streams the next frame's
    workgroups start on the CUs the previous frame has left: one rank's 80 000-ray shard of eight 0.658 -> 0.627 ms per pipelined step
    (profiles/r06_two_streams.txt).  A field handle supports overlapping launches (one tile-queue slot per launch, 256 in flight).
    The pipeline's five streams (these two, RCCL's, the reorder's, the caller's) need hardware queues of their own: the HIP runtime maps a process's streams
    onto FOUR unless GPU_MAX_HW_QUEUES says otherwise (read when the runtime starts), and two streams that share a queue run one behind the other -- one
    rank's pipelined step beside an exchange that holds CUs: 0.64-0.69 ms on four queues, 0.61-0.65 on eight (profiles/r06_exchange_contention.txt).
    Set GPU_MAX_HW_QUEUES=8 in the environment of a multi-GPU job (bench.py does)."""
    cur = torch.cuda.current_stream(device)
    out = [torch.cuda.Stream(device) for _ in range(n)]
    for s in out:
        s.wait_stream(cur)
    return out


class PipelinedGather:
    """Double-buffered all-gather of a rank's pixels: frame k's exchange runs on RCCL's stream while frame k+1 is
    marched on the render stream.  ``buffers(k)`` gives the (rgb, depth) views frame k must be rendered into;
    ``submit(k)`` starts its exchange; ``frame(k)`` waits for it and returns the rank-major (rgb, depth)."""

    def __init__(self, per: int, world: int, device, group=None, depth: int = 2):
        self.per, self.world, self.group = per, world, group
        self.send = [shard_buffers(per, device) for _ in range(depth)]
        self.recv = [torch.empty((world * 4 * per,), device=device, dtype=torch.float32) for _ in range(depth)]
        self.work = [None] * depth
        self.reorder_done = [None] * depth      # events: the side-stream copy that last READ receive buffer i (frame_in_image_order(stream=...))

    def buffers(self, k: int):
        i = k % len(self.send)
        if self.work[i] is not None:          # the exchange that last read this send buffer must be done
            self.work[i].wait()
            self.work[i] = None
        if self.reorder_done[i] is not None:  # ... also when a side stream did the waiting (frame_in_image_order(stream=...)): its copy ran behind that exchange
            torch.cuda.current_stream().wait_event(self.reorder_done[i])
            self.reorder_done[i] = None       # everything enqueued on this stream from here on (the render, the exchange ``submit`` starts) is behind that copy
        _, rgb, depth = self.send[i]
        return rgb, depth

    def submit(self, k: int):
        i = k % len(self.send)
        if self.reorder_done[i] is not None:      # the exchange overwrites receive buffer i: the side-stream copy of frame k - depth must have read it
            torch.cuda.current_stream().wait_event(self.reorder_done[i])
            self.reorder_done[i] = None
        self.work[i] = dist.all_gather_into_tensor(self.recv[i], self.send[i][0], group=self.group, async_op=True)

    def frame(self, k: int):
        i = k % len(self.send)
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        blocks = self.recv[i].view(self.world, 4 * self.per)
        return blocks[:, : 3 * self.per].reshape(self.world * self.per, 3), blocks[:, 3 * self.per:].reshape(self.world * self.per)

    def frame_in_image_order(self, k: int, H: int, W: int, block: int, out=None, stream=None):
        """Frame k in IMAGE order, straight from the receive buffer: ONE strided copy per output (rgb, depth).  ``frame(k)`` followed by
        ``deinterleave`` is four -- slicing the rank-major [world, 4 per] buffer into rgb and depth makes a packed copy, the row permutation
        another -- on the stream the next frame's render is waiting on; at eight ranks a 0.64 ms step has 0.08 ms to spare.
        ``out`` = (rgb [H*W,3], depth [H*W]) to write into (else new tensors).  ``stream`` (CUDA tensors): wait for the exchange and copy on THAT stream
        instead of the caller's -- the next frame's render, enqueued on the caller's stream, then waits for neither; ``submit`` makes the exchange that
        reuses the receive buffer wait for the copy.  The caller synchronises with ``stream`` before it reads ``out`` (bench.py: the device-wide
        synchronise that ends the timed region)."""
        i = k % len(self.send)
        if H % (self.world * block) != 0 or (H // (self.world * block)) * block * W * self.world != self.world * self.per:
            raise ValueError(f"frame_in_image_order: {H} rows do not split into blocks of {block} rows dealt out to {self.world} ranks "
                             f"({self.per} pixels per rank, width {W}): H must be a multiple of world x block and per = H W / world")
        if stream is not None and self.recv[i].is_cuda:
            if out is None:
                # tensors allocated inside the side-stream context would be handed to the caller's stream with no ordering and no record_stream
                # (the caching allocator could reuse their memory while the copy still runs): the caller owns the output buffers
                raise ValueError("frame_in_image_order(stream=...) writes into caller-owned buffers: pass out=(rgb [H*W,3], depth [H*W])")
            with torch.cuda.stream(stream):
                res = self.frame_in_image_order(k, H, W, block, out=out)
                ev = torch.cuda.Event()
                ev.record(stream)
            self.reorder_done[i] = ev
            return res
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        world, per = self.world, self.per
        nblk = H // (world * block)
        blocks = self.recv[i].view(world, 4 * per)
        rgb_v = blocks[:, : 3 * per].view(world, nblk, block * W * 3).permute(1, 0, 2)           # [nblk, world, block*W*3]: image order, strided
        dep_v = blocks[:, 3 * per:].view(world, nblk, block * W).permute(1, 0, 2)
        if out is None:
            out = (torch.empty((H * W, 3), device=blocks.device, dtype=blocks.dtype), torch.empty((H * W,), device=blocks.device, dtype=blocks.dtype))
        out[0].view(nblk, world, block * W * 3).copy_(rgb_v)
        out[1].view(nblk, world, block * W).copy_(dep_v)
        return out

    def drain(self):
        for i, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[i] = None


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
