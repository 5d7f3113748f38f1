#!/usr/bin/env python3
"""Is a launch's time a function of WHERE its ray list and output buffers live?  (profiles/exp_rank_shards.py times one of the two N = 2 shards 4-5 % slower than the
other although both cost the same when all tensors are made up front.)  The 320 000-ray shard of rank 1, rays / outputs carved out of one big buffer at a sweep of byte offsets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ngf_amd  # noqa: F401
from ngf_amd import cases, dist, synth
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device="cuda", bake=True, bake_color=True)
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda().view(800, 800, 6)
rows = dist.interleaved_rows(800, 2, 1, 10)
src = torch.cat([frame[a:b] for a, b in rows]).reshape(-1, 6).contiguous()
n = src.shape[0]
big = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)          # 64 MiB arena
base = big.data_ptr()
print(f"arena at {base:#x}; rays {n * 24} B, rgb {n * 12} B, depth {n * 4} B")
def carve(off, count):
    return big[off: off + 4 * count].view(torch.float32)
def timed(rays, out, rep=12):
    for _ in range(2): f(rays, N_samples=192, white_bg=True, iteration=30001, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rep)]
    for a, b in ev:
        a.record(); f(rays, N_samples=192, white_bg=True, iteration=30001, out=out); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
MB = 1 << 20
for label, roff in (("0", 0), ("256 B", 256), ("4 KiB", 4096), ("64 KiB", 65536), ("1 MiB", MB), ("2 MiB", 2 * MB), ("2 MiB + 4 KiB", 2 * MB + 4096), ("3 MiB", 3 * MB)):
    rays = carve(roff, n * 6).view(n, 6); rays.copy_(src)
    o1 = 16 * MB
    rgb = carve(o1, n * 3).view(n, 3); dep = carve(o1 + 8 * MB, n)
    print(f"rays at +{label:14s}: {timed(rays, (rgb, dep)):.3f} ms")
rays = carve(0, n * 6).view(n, 6); rays.copy_(src)
for label, ooff in (("0", 0), ("256 B", 256), ("4 KiB", 4096), ("1 MiB", MB), ("1 MiB + 64 KiB", MB + 65536)):
    rgb = carve(16 * MB + ooff, n * 3).view(n, 3); dep = carve(24 * MB + ooff, n)
    print(f"outputs at +{label:14s}: {timed(rays, (rgb, dep)):.3f} ms")
# the allocator's own placements, as exp_rank_shards.py gets them: fresh tensors, no `out`
for trial in range(4):
    r = src.clone()
    def t2():
        for _ in range(2): f(r, N_samples=192, white_bg=True, iteration=30001)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for a, b in ev:
            a.record(); o = f(r, N_samples=192, white_bg=True, iteration=30001); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev])), o
    ms, o = t2()
    print(f"fresh tensors, trial {trial}: rays {r.data_ptr() - base:+#x} rgb {o['rgb_map'].data_ptr() - base:+#x} depth {o['depth_map'].data_ptr() - base:+#x}: {ms:.3f} ms")
    keep = torch.empty((trial + 1) * 3_000_000, device="cuda")       # shift the allocator's next placements
