import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ngf_amd
from ngf_amd import _lib, cases, dist, synth
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device="cuda", bake=True, bake_color=True)
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda().view(800, 800, 6)
def timed(rays, rep=20, **kw):
    for _ in range(3): f(rays, N_samples=192, white_bg=True, iteration=30001, **kw)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rep)]
    for a, b in ev:
        a.record(); f(rays, N_samples=192, white_bg=True, iteration=30001, **kw); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
shards = {}
for world in (2,):
    for rank in range(world):
        rows = dist.interleaved_rows(800, world, rank, 10)
        shards[rank] = torch.cat([frame[a:b] for a, b in rows]).reshape(-1, 6).contiguous()
for order in ((0, 1), (1, 0), (0, 1), (1, 0)):
    print("order", order, " ".join(f"rank{r}: {timed(shards[r]):.3f} (row_width: {timed(shards[r], row_width=800):.3f})" for r in order))
# other block sizes / contiguous halves
for blk in (10, 20, 40, 50, 100, 400):
    ms = []
    for rank in range(2):
        rows = dist.interleaved_rows(800, 2, rank, blk)
        r = torch.cat([frame[a:b] for a, b in rows]).reshape(-1, 6).contiguous()
        ms.append((timed(r), timed(r, row_width=800)))
    print(f"block {blk}: " + "  ".join(f"{a:.3f} ({b:.3f})" for a, b in ms))
# stats per shard
for r in (0, 1):
    f(shards[r], N_samples=192, white_bg=True, iteration=30001, collect_stats=True)
    print("rank", r, "stats", f.last_stats.cpu().numpy()[:4])
