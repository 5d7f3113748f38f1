"""Host-side logic on CPU: config parser (flag surface of TriPlane/opt.py), synthesiser determinism,
ray generation, and the ray-sharded render over a world_size-2 gloo group."""
import os
import sys

import numpy as np
import pytest
import torch

import ngf_amd  # noqa: F401
from ngf_amd import dist as ndist
from ngf_amd import opt, synth

LEGO_TXT = """
model_name = TriPlane
expname = TriPlane/lego
datadir = /data/NeRF-Synthetic/lego
basedir = ./log
dataset_name = blender

n_iters = 30000
batch_size = 4096

N_voxel_init = 16777216 #256**3
N_voxel_final = 27000000 # 300**3
upsamp_list = [2000, 2500] #[2000,3000,4000,5500,7000]
update_AlphaMask_list = [2000, 2500]

N_vis = 5
vis_every = 2100 # 30001
render_test = 1
gauge_start=4000
"""


def test_config_parser_defaults_and_config_file(tmp_path):
    a = opt.config_parser([])
    assert (a.model_name, a.batch_size, a.distance_scale, a.step_ratio, a.gauge_start) == ("TensorVMSplit", 4096, 25, 0.5, 0)
    assert a.alpha_mask_thre == 1e-4 and a.nSamples == 1e6 and a.N_voxel_init == 100 ** 3 and a.upsamp_list is None
    assert a.white_bkgd is False and a.dataset_name == "blender" and a.basedir == "./log"
    cfg = tmp_path / "lego.txt"
    cfg.write_text(LEGO_TXT)
    b = opt.config_parser(["--config", str(cfg), "--batch_size", "8192"])
    assert b.model_name == "TriPlane" and b.N_voxel_init == 256 ** 3 and b.N_voxel_final == 27000000
    assert b.upsamp_list == [2000, 2500] and b.update_AlphaMask_list == [2000, 2500]
    assert b.gauge_start == 4000 and b.vis_every == 2100 and b.render_test == 1
    assert b.batch_size == 8192                                  # command line overrides the file
    c = opt.config_parser(["--infoinv"], infoinv=True)
    assert c.infoinv is True and not hasattr(c, "gauge_start")
    assert opt.config_parser([], infoinv=True).infoinv is False   # store_true, default False (InfoInv/opt.py:117)
    with pytest.raises(SystemExit):
        opt.config_parser(["--dataset_name", "nope"])


def test_synth_is_bit_reproducible():
    a = synth.hash_normal(5, 3, (1000,))
    assert a.dtype == np.float32 and abs(float(a.std()) - 1.0) < 0.05
    # pinned values: any platform must regenerate these exactly
    assert np.array_equal(synth.hash_uniform(1, 2, (3,)), synth.hash_uniform(1, 2, (3,)))
    p = synth.triplane_params(7, ((6, 8),) * 3, (4, 4), preset="R1")
    q = synth.triplane_params(7, ((6, 8),) * 3, (4, 4), preset="R1")
    assert all(np.array_equal(p[k], q[k]) for k in p)
    assert p["density_decoder.bias"][0] == 10 and p["plane_xy"].shape == (1, 64, 6, 8)


def test_lookat_rays_geometry():
    r = synth.lookat_rays(800, 800, rows=(400, 401))
    assert r.shape == (800, 6)
    np.testing.assert_allclose(np.linalg.norm(r[:, 3:], axis=1), 1.0, atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(r[0, :3]), 4.0311, rtol=1e-6)
    # central pixel looks (almost) at the origin
    c = r[400]
    t = -np.dot(c[:3], c[3:])
    assert np.linalg.norm(c[:3] + t * c[3:]) < 0.01
    full = synth.lookat_rays(16, 16)
    part = synth.lookat_rays(16, 16, rows=(4, 9))
    assert np.array_equal(full[4 * 16:9 * 16], part)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from helpers import load_case, oracle_for_case
    g, params, step, mask = load_case("triplane_r1_gauge")
    orc = oracle_for_case(g, params, step, mask)          # the ORACLE stands in for the device here (test only)
    rays = torch.from_numpy(g["rays"][:201])              # not divisible by 2: exercises the padded shard

    def render_fn(shard):
        rgb, depth = orc.render(shard.numpy(), 48, threads=1)
        return torch.from_numpy(rgb), torch.from_numpy(depth)

    rgb, depth = ndist.render_sharded(render_fn, rays)
    ref_rgb, ref_depth = orc.render(rays.numpy(), 48, threads=1)
    ok = np.array_equal(rgb.numpy(), ref_rgb) and np.array_equal(depth.numpy(), ref_depth)
    lo, hi, per = ndist.shard_bounds(201, world, rank)
    ok = ok and (per == 101) and (hi - lo == (101 if rank == 0 else 100))
    # interleaved row blocks + double-buffered exchange (what bench.py runs at N > 1): three 8x5 "frames"
    H, W, blk = 8, 5, 2
    frames = [rays[k:k + H * W] for k in (0, 40, 80)]
    full = [orc.render(fr.numpy(), 48, threads=1) for fr in frames]
    rows = ndist.interleaved_rows(H, world, rank, blk)
    ok = ok and rows == [(rank * blk, rank * blk + blk), ((rank + 2) * blk, (rank + 2) * blk + blk)]
    pipe = ndist.PipelinedGather(H * W // world, world, torch.device("cpu"))
    got = []
    for k, fr in enumerate(frames):
        mine = torch.cat([fr[r0 * W:r1 * W] for r0, r1 in rows])
        o_rgb, o_depth = pipe.buffers(k)
        r, d = render_fn(mine)
        o_rgb.copy_(r); o_depth.copy_(d)
        pipe.submit(k)
        if k > 0:
            got.append(ndist.deinterleave(*pipe.frame(k - 1), H, W, world, blk))
    got.append(pipe.frame_in_image_order(len(frames) - 1, H, W, blk))          # the fused form bench.py uses: one strided copy per output
    for (g_rgb, g_depth), (f_rgb, f_depth) in zip(got, full):
        ok = ok and np.array_equal(g_rgb.numpy(), f_rgb) and np.array_equal(g_depth.numpy(), f_depth)
    open(os.path.join(tmp, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_sharded_render_world2_gloo(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(2)] == ["1", "1"]


def _worker_rows(rank, world, port, tmp):
    """The bench's data path at world 4 / 8 without a device: every rank fills its interleaved 10-row blocks of an 800 x 800 frame with
    a value that encodes (row, column), the exchange is PipelinedGather's double-buffered all_gather (gloo), and the re-ordered frames
    must be the row-major frame -- for depth + 2 frames in flight, i.e. every buffer of the pipeline is reused (depth ndist.PIPELINE_DEPTH = what bench.py runs since its frames
    alternate between two render streams, depth 2 = the class default)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H = W = 800
    blk = 10
    rows = ndist.interleaved_rows(H, world, rank, blk)
    per = H * W // world
    ok = sum(r1 - r0 for r0, r1 in rows) * W == per and len(rows) == H // (world * blk)
    pipe = ndist.PipelinedGather(per, world, torch.device("cpu"), depth=ndist.PIPELINE_DEPTH if world == 4 else 2)
    ok = ok and pipe.send[0][0].numel() == 4 * per and pipe.recv[0].numel() == world * 4 * per          # one [per,4] operand per rank
    got = []

    def fill(k):
        rr = torch.cat([torch.arange(r0, r1) for r0, r1 in rows]).view(-1, 1).expand(-1, W).reshape(-1)
        cc = torch.arange(W).repeat(rr.numel() // W)
        base = (rr * W + cc).to(torch.float32)                     # < 2^24: exact in float32
        o_rgb, o_depth = pipe.buffers(k)
        o_rgb.copy_(torch.stack([base, base + 0.25, base + 0.5], 1) + 1000000.0 * k)
        o_depth.copy_(base + 0.75 + 1000000.0 * k)

    nfr = len(pipe.send) + 2
    for k in range(nfr):
        fill(k)
        pipe.submit(k)
        if k > 0:
            got.append(pipe.frame_in_image_order(k - 1, H, W, blk))          # what bench.py calls per step
    got.append(ndist.deinterleave(*pipe.frame(nfr - 1), H, W, world, blk))          # and the two-step form: the same frame
    pix = torch.arange(H * W, dtype=torch.float32)
    for k, (g_rgb, g_depth) in enumerate(got):
        ok = ok and torch.equal(g_rgb[:, 0], pix + 1000000.0 * k) and torch.equal(g_rgb[:, 2], pix + 0.5 + 1000000.0 * k)
        ok = ok and torch.equal(g_depth, pix + 0.75 + 1000000.0 * k)
    open(os.path.join(tmp, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_interleaved_row_exchange_world_4_and_8_gloo(tmp_path, world):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_rows, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(world)] == ["1"] * world


def test_interleaved_rows_partition_the_frame():
    """interleaved_rows for every world size bench.py accepts: the ranks' row sets are disjoint, cover the 800 rows, have equal
    size, and deinterleave is the inverse permutation of the rank-major layout."""
    H, W, blk = 800, 800, 10
    for world in (1, 2, 4, 8):
        sets = [ndist.interleaved_rows(H, world, r, blk) for r in range(world)]
        rows = sorted(r for s_ in sets for (r0, r1) in s_ for r in range(r0, r1))
        assert rows == list(range(H)) and len({sum(r1 - r0 for r0, r1 in s_) for s_ in sets}) == 1
        order = torch.cat([torch.arange(r0 * W, r1 * W) for s_ in sets for (r0, r1) in s_]).to(torch.float32)      # rank-major pixel ids
        rgb = torch.stack([order, order, order], 1)
        back_rgb, back_depth = ndist.deinterleave(rgb, order.clone(), H, W, world, blk)
        assert torch.equal(back_depth, torch.arange(H * W, dtype=torch.float32)) and torch.equal(back_rgb[:, 1], back_depth)
    with pytest.raises(ValueError):
        ndist.interleaved_rows(800, 3, 0, 10)


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 640000):
        for w in (1, 2, 4, 8):
            spans = [ndist.shard_bounds(n, w, r)[:2] for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_ray_restatements_match_reference_fixtures():
    """The numpy restatements of the loaders' ray generation (synth.lookat_rays, synth.dtu_rays_dir: what the sharded bench and
    the CPU-side checks feed on) against rays produced by the reference's own functions (make_golden.capture_rays)."""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = np.load(os.path.join(here, "rays_blender.npz"))
    for tag, angle in (("a", 0.6911112070083618), ("b", 0.9)):
        H, W, c2w = int(fx[tag + "_H"]), int(fx[tag + "_W"]), fx[tag + "_c2w"]
        for k, r in enumerate(fx[tag + "_rows"]):
            mine = synth.lookat_rays(H, W, c2w, camera_angle_x=angle, rows=(int(r), int(r) + 1))
            assert np.array_equal(mine[:, :3], fx[tag + "_rays"][k][:, :3])
            assert np.abs(mine[:, 3:] - fx[tag + "_rays"][k][:, 3:]).max() <= 3e-7        # matmul rounding order
    fd = np.load(os.path.join(here, "rays_dtu.npz"))
    for v in (0, 33):
        for k, r in enumerate(fd[f"v{v}_rows"]):
            mine = synth.dtu_rays_dir(600, 800, fd[f"v{v}_focal"], fd[f"v{v}_princpt"], fd[f"v{v}_rot"], rows=(int(r), int(r) + 1))
            assert np.array_equal(mine, fd[f"v{v}_raydir"][k])


def test_optimisation_level_flags_resolve_in_the_constructor():
    """bake_density=None (the default) means level 2 unless level 0 (no_fold) is asked for: TriPlane(..., no_fold=True) alone must not end in
    ngf_field_create's NGF_E_ARG (NGF_F_NO_FOLD excludes the NGF_F_BAKE_* flags)."""
    from ngf_amd import triplane
    aabb = torch.tensor([[-1.5] * 3, [1.5] * 3])
    mk = lambda **kw: triplane.TriPlane(aabb, [16, 16, 16], "cpu", **kw)
    f = mk()
    assert f.bake_density and f.bake_color and not f.no_fold                     # level 3: the default since round 4
    f = mk(bake_color=False)
    assert f.bake_density and not f.bake_color                                   # level 2
    f = mk(no_fold=True)
    assert f.no_fold and not f.bake_density and not f.bake_color                 # level 0
    f = mk(bake_density=False)
    assert not f.bake_density and not f.bake_color and not f.no_fold             # level 1
    f = mk(split_bf16=True)
    assert f.bake_density and not f.bake_color and f.split_bf16                  # the bf16 split works on level 2
    f = mk(bake_color=True)
    assert f.bake_density and f.bake_color                                       # level 3
    for bad in (dict(no_fold=True, bake_density=True), dict(no_fold=True, bake_color=True)):
        with pytest.raises(ValueError):
            mk(**bad)


def test_tile_plan_covers_every_ray_once_with_whole_tiles():
    """launch_render's tile plan (wide tiles first, narrow tiles for the last rays so that the persistent grid runs dry together): host
    arithmetic, checked here for every launch size class -- segments in ray order, widths strictly decreasing, every segment but the last a
    whole number of tiles, all rays covered exactly once; and the documented shapes (a frame keeps >= 95 % of its rays in the widest tiles,
    a launch smaller than the resident grid is all one-ray tiles, tail = 0 is the widest tile only)."""
    import ctypes as C
    from ngf_amd import _lib
    L = _lib.lib()

    def plan(n, wide, resident, tail):
        r, sh = (C.c_int64 * 4)(), (C.c_int32 * 4)()
        k = L.ngf_debug_tile_plan(n, wide, resident, tail, r, sh)
        assert 1 <= k <= 4
        return [(int(r[i]), 1 << int(sh[i])) for i in range(k)]

    rng = np.random.default_rng(5)
    sizes = [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 4096, 40000, 80000, 160000, 640000, 640001, 2 ** 31 + 11] + [int(v) for v in rng.integers(1, 3_000_000, 200)]
    for wide in (8, 16):
        for resident in (12, 24, 3072, 2048):
            for tail in (-1, 0, 1, 8, 16, 40):
                for n in sizes:
                    p = plan(n, wide, resident, tail)
                    assert sum(r for r, _ in p) == n, (n, wide, resident, tail, p)
                    assert all(r > 0 for r, _ in p)
                    assert all(a[1] > b[1] for a, b in zip(p, p[1:])), p              # widths strictly decreasing
                    assert all(r % w == 0 for r, w in p[:-1]), p                      # whole tiles everywhere but at the end of the list
                    assert all(w in (wide, 4, 2, 1) for _, w in p)
    p = plan(640000, 8, 3072, -1)
    assert p[0][1] == 8 and p[0][0] >= 0.95 * 640000 and p[-1] == (3072, 1)
    assert plan(2000, 8, 3072, -1) == [(2000, 1)]
    # round 5: a launch that fits the grid with one tile per wave takes the narrowest pair of widths that does it -- never a second tile behind the first
    assert plan(4096, 8, 3072, -1) == [(2048, 2), (2048, 1)] and plan(3073, 8, 3072, -1) == [(2, 2), (3071, 1)]
    assert plan(8000, 8, 3072, -1) == [(3712, 4), (4288, 2)] and plan(6144, 8, 3072, -1) == [(6144, 2)]
    for n in (3072, 4096, 5000, 6144, 9000, 12288, 20000, 24576):
        p = plan(n, 8, 3072, -1)
        assert sum((r + w - 1) // w for r, w in p) <= 3072, (n, p)
    assert plan(640000, 8, 3072, 0) == [(640000, 8)]
    assert plan(80000, 8, 3072, 16) == [(58496, 8), (12288, 4), (6144, 2), (3072, 1)]


def test_screen_space_tile_order_is_a_bijection():
    """ngf_field_render_image re-orders the queue positions of the widest plan segment into an image-blocked walk (RenderArgs::ord_*): for every
    plan -- block widths that do not divide the row, bands of every height, a tail the map leaves alone -- the map is a bijection of the tiles,
    the identity behind ord_n, and consecutive positions of one block stay inside that block's bh x bw window."""
    import ctypes as C
    from ngf_amd import _lib
    L = _lib.lib()
    for tpr, bw, bh, rows, extra in ((100, 10, 80, 800, 0), (100, 10, 80, 790, 37), (53, 10, 7, 40, 5), (7, 3, 2, 9, 0), (10, 10, 4, 8, 3), (25, 4, 80, 3, 0), (1, 1, 2, 16, 0)):
        bh_eff = min(bh, rows)
        ord_n = (rows // bh_eff) * bh_eff * tpr
        total = rows * tpr + extra
        q = np.arange(total, dtype=np.uint32)
        out = np.empty_like(q)
        assert L.ngf_debug_tile_order(q.ctypes.data_as(C.c_void_p), total, ord_n, tpr, bw, bh_eff, out.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(np.sort(out[:ord_n]), np.arange(ord_n, dtype=np.uint32)), (tpr, bw, bh_eff)
        assert np.array_equal(out[ord_n:], q[ord_n:])
        r, c = out[:ord_n] // tpr, out[:ord_n] % tpr
        # the first block of the first band: its bw x bh positions cover exactly columns [0, bw) of rows [0, bh)
        w0 = min(bw, tpr)
        first = slice(0, w0 * bh_eff)
        if ord_n >= w0 * bh_eff:
            assert r[first].max() == bh_eff - 1 and c[first].max() == w0 - 1 and len(set(zip(r[first].tolist(), c[first].tolist()))) == w0 * bh_eff
        # a band never leaves its rows
        band = np.arange(ord_n) // (tpr * bh_eff)
        assert np.array_equal(r // bh_eff, band)
