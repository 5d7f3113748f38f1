#!/usr/bin/env python3
"""What does an exchange kernel that NEEDS CUs do to one rank's pipeline of `bench.py --gpus 8`?  No multi-GPU node has run the bench, and on one GPU the
world-1 RCCL all-gather is a local copy: it never competes with the render for CUs.  On a real node RCCL's kernel (~100 registers per lane) cannot share a
CU with the level-3 render workgroup (12 waves x 168 registers), so it is placed only where a render workgroup has LEFT -- while the next frame's persistent
grid (the other render stream) waits for the same CUs -- and it keeps its CUs while it waits for its peers.  Stand-in: profiles/micro/cu_holder.hip, `wgs`
workgroups of 256 threads x 128 registers that hold their CU slots for `us` microseconds, enqueued behind every frame's render on a stream of its own
(normal or high priority, as bench.py's RCCL stream), followed by the 10 MB reorder copy on the side stream; the march of frame k waits for the reorder of
frame k - depth, exactly as PipelinedGather.buffers does.  Reported: ms per pipelined step (80 000-ray shard, two alternating render streams) and how long
after the END of frame k's render its exchange ends (median / max) -- the latency the pipeline depth has to cover.
    hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o profiles/micro/libcu_holder.so profiles/micro/cu_holder.hip
    python profiles/exp_exchange_contention.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
if os.environ.get("LATE_QUEUES"):          # does the HIP runtime still read GPU_MAX_HW_QUEUES after torch has been imported (before the first torch.cuda call)?
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["LATE_QUEUES"]
import ngf_amd  # noqa: F401
from ngf_amd import cases, dist as ndist, synth

dev = torch.device("cuda:0")
H = C.CDLL(os.path.join(ROOT, "profiles", "micro", "libcu_holder.so"))
H.cu_holder_launch.argtypes = [C.c_int, C.c_longlong, C.c_void_p]
WORLD = int(os.environ.get("SHARDS", "8"))
g, params, step = cases.big_case("triplane", "R1")
f = cases.field_for_case(g, params, None, device="cuda", bake=True, bake_color=True)
frame = torch.from_numpy(synth.lookat_rays(800, 800)).cuda().view(800, 800, 6)
rows = ndist.interleaved_rows(800, WORLD, 0, 10)
rays = torch.cat([frame[a:b] for a, b in rows]).reshape(-1, 6).contiguous()
per = rays.shape[0]
MAXD = 8
out = [(torch.empty((per, 3), device=dev), torch.empty((per,), device=dev)) for _ in range(MAXD)]
big_src, big_dst = torch.empty(2_560_000, device=dev), torch.empty(2_560_000, device=dev)      # 10 MB: the reorder of an eight-rank frame
rstreams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
side = torch.cuda.Stream(dev)
comm = {"normal": torch.cuda.Stream(dev), "high": torch.cuda.Stream(dev, priority=-1)}


def run(steps, wgs, us, prio, depth, nstreams=2, mode="", bufs=None, side_copy=True):
    cs = comm[prio]
    ev_r = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ev_c = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    reorder_done = [None] * depth
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        st = rstreams[k % nstreams]
        i = k % depth
        ib = k % (bufs or depth)
        with torch.cuda.stream(st):
            if reorder_done[i] is not None:
                st.wait_event(reorder_done[i])
                reorder_done[i] = None
            if mode == "stagger0" and k == 1:
                st.wait_event(ev_r[0])          # the pipeline starts from idle: the second frame starts when the first has ended, every later one in its predecessor's tail
            if mode == "couple3" and k >= 3:
                st.wait_event(ev_r[k - 3])      # the coupling an odd pipeline depth has, without the exchange in it
            f(rays, N_samples=192, white_bg=True, out=out[ib], iteration=30001, row_width=800)
            ev_r[k].record(st)
        cs.wait_event(ev_r[k])
        if wgs > 0:
            H.cu_holder_launch(wgs, int(us * 1000), C.c_void_p(cs.cuda_stream))
        ev_c[k].record(cs)
        if k > 0:
            with torch.cuda.stream(side):
                side.wait_event(ev_c[k - 1])
                if side_copy:
                    big_dst.copy_(big_src)
                e = torch.cuda.Event()
                e.record(side)
            reorder_done[(k - 1) % depth] = e
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps * 1e3
    lag = np.array([ev_r[k].elapsed_time(ev_c[k]) for k in range(5, steps - 5)])
    return el, float(np.median(lag)), float(lag.max())


for _ in range(40):
    f(rays, N_samples=192, white_bg=True, out=out[0], iteration=30001, row_width=800)
torch.cuda.synchronize()
print(f"{per} rays per step (rank 0 of {WORLD}), level 3, two alternating render streams; 5 runs of 100 steps, median ms per step | exchange ends after its render's end: median / max ms")
run(100, 0, 0, "normal", 4)
if os.environ.get("DEPTHS"):
    # the pipeline depth alone (interleaved repetitions: drift between blocks does not sort by depth)
    depths = [int(x) for x in os.environ["DEPTHS"].split(",")]
    res = {(d, w): [] for d in depths for w in (0, 32)}
    for rep in range(4):
        for d in depths:
            for w in (0, 32):
                res[(d, w)].append(run(100, w, 150, "high", d))
    for d in depths:
        print(f"depth {d}:  no exchange kernel {np.median([r[0] for r in res[(d, 0)]]):.4f} ms per step   32 workgroups x 150 us {np.median([r[0] for r in res[(d, 32)]]):.4f}"
              f" | exchange ends {np.median([r[1] for r in res[(d, 32)]]):.3f} / {max(r[2] for r in res[(d, 32)]):.3f} ms after its render", flush=True)
    sys.exit(0)
if os.environ.get("MODES"):
    cfgs = [(d, m) for d in (3, 4) for m in ("", "stagger0", "couple3")]
    res = {(c, w): [] for c in cfgs for w in ((0, 0), (16, 50), (32, 150))}
    for rep in range(4):
        for c in cfgs:
            for w in ((0, 0), (16, 50), (32, 150)):
                res[(c, w)].append(run(100, w[0], w[1], "high", c[0], mode=c[1]))
    for c in cfgs:
        print(f"depth {c[0]} {c[1] or 'as is':9s}: " + "   ".join(f"{w[0]:2d} x {w[1]:3d} us {np.median([r[0] for r in res[(c, w)]]):.4f}" for w in ((0, 0), (16, 50), (32, 150))), flush=True)
    sys.exit(0)
if os.environ.get("SPLIT"):
    # which of the two things the depth sets makes the parity effect: the output buffers a stream cycles through, or the distance of the reorder a frame waits for?
    cfgs = [(3, 3, True), (4, 4, True), (3, 4, True), (4, 3, True), (3, 1, True), (4, 1, True), (100, 3, True), (100, 4, True), (3, 3, False), (4, 4, False)]
    res = {(c, w): [] for c in cfgs for w in ((0, 0), (32, 150))}
    for rep in range(4):
        for c in cfgs:
            for w in ((0, 0), (32, 150)):
                res[(c, w)].append(run(100, w[0], w[1], "high", c[0], bufs=c[1], side_copy=c[2]))
    for c in cfgs:
        print(f"wait for the reorder of frame k - {c[0]:3d}, {c[1]} output buffers, side copy {c[2]}: " + "   ".join(f"{w[0]:2d} x {w[1]:3d} us {np.median([r[0] for r in res[(c, w)]]):.4f}" for w in ((0, 0), (32, 150))), flush=True)
    sys.exit(0)
if os.environ.get("STREAMS"):
    # one render stream against two alternating ones, with an exchange that needs CUs
    cfgs = [(ns, d) for ns in (1, 2) for d in (2, 3, 4, 6)]
    loads = ((0, 0), (16, 50), (32, 150), (32, 400))
    res = {(c, w): [] for c in cfgs for w in loads}
    for rep in range(4):
        for c in cfgs:
            for w in loads:
                res[(c, w)].append(run(100, w[0], w[1], "high", c[1], nstreams=c[0]))
    for c in cfgs:
        print(f"{c[0]} render stream(s), depth {c[1]}: " + "   ".join(f"{w[0]:2d} x {w[1]:3d} us {np.median([r[0] for r in res[(c, w)]]):.4f}" for w in loads), flush=True)
    sys.exit(0)
if os.environ.get("CAPPED"):
    # eight held CUs (RCCL capped at eight channels, as bench.py sets it): one render stream against two, four buffers
    loads = ((0, 0), (8, 50), (8, 150), (8, 400), (8, 1000))
    cfgs = [(1, 4), (2, 4), (1, 3), (2, 3)]
    res = {(c, w): [] for c in cfgs for w in loads}
    for rep in range(4):
        for c in cfgs:
            for w in loads:
                res[(c, w)].append(run(100, w[0], w[1], "high", c[1], nstreams=c[0]))
    for c in cfgs:
        print(f"{c[0]} render stream(s), depth {c[1]}: " + "   ".join(f"{w[0]:2d} x {w[1]:4d} us {np.median([r[0] for r in res[(c, w)]]):.4f}" for w in loads), flush=True)
    sys.exit(0)
if os.environ.get("QUICK"):
    rs = [run(100, 32, 400, "high", 4) for _ in range(5)]
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} (late: {os.environ.get('LATE_QUEUES')}): two streams, depth 4, 32 x 400 us: {np.median([r[0] for r in rs]):.4f} ms per step", flush=True)
    sys.exit(0)
for depth in (4, 3, 2):
    for wgs, us in ((0, 0), (16, 50), (16, 150), (32, 50), (32, 150), (64, 150), (32, 400)):
        line = f"depth {depth}  exchange {wgs:3d} workgroups x {us:3d} us: "
        for prio in ("normal", "high"):
            rs = [run(100, wgs, us, prio, depth) for _ in range(5)]
            line += f"  {prio:6s} priority {np.median([r[0] for r in rs]):.4f} | {np.median([r[1] for r in rs]):.3f} / {max(r[2] for r in rs):.3f}"
        print(line, flush=True)
