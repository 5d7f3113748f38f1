"""UV-Mapping (NeuTex) colour path on the GPU against the C oracle and the golden vectors captured from the
reference's sub-modules.  The path is ill-conditioned by construction (PE with 2^9 on positions and on uv feeds an
11-layer MLP): a 1-ulp change of a sample position moves a sample colour by ~1e-3.  Per-sample quantities are
therefore compared with a looser bound than the TriPlane path; the composited pixels are held to PIX_TOL = 2e-5 against the oracle AND
the reference's golden pixels (north_star: 1e-4; measured 4e-6 -- a 5x regression fails here)."""
import numpy as np
import pytest

from helpers import load_uv_case
from oracle.oracle import OracleUV

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

PIX_TOL = 2e-5          # composited pixels vs oracle / reference golden
T_TOL = 1e-5            # transmittance
SIGMA_RTOL = 2e-5       # per-sample density (relative); measured 1.5e-6 ... 2.2e-6 in both kernels (round 5: was 5e-4)
PCOL_TOL = 1e-3         # per-sample colour (absolute): the ill-conditioned quantity; measured 1.8e-4 (sphere) / 4.8e-5 (square) (round 5: was 5e-3)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("name", ["uv_sphere", "uv_square"])
def test_uv_matches_oracle_and_reference(name, split):
    """Per-sample density AND composited pixels, fp32 kernel and NGF_UV_F_SPLIT_BF16 (round 5: both modes -- the one intermediate build of round 4 with
    wrong densities, DESIGN.md section 6.7, was wrong in ONE of the two instantiations only; profiles/exp_uv_split_check.py is this check as a script)."""
    from ngf_amd import uvmapping
    g, params = load_uv_case(name)
    pt = str(g["primitive_type"])
    orc = OracleUV(params, pt)
    o_color, o_trans, dbg = orc.render(g["campos"], g["raydir"], g["U"], bg=g["bg"], debug=True)
    m = uvmapping.NeuTex(primitive_type=pt, sample_num=int(g["S"]), device="cuda", split_bf16=split)
    m.load_params(params)
    out = m(torch.from_numpy(g["campos"])[None], torch.from_numpy(g["raydir"])[None], torch.from_numpy(g["bg"])[None],
            jitter_u=torch.from_numpy(g["U"])[None], debug=True)
    color, trans = out["color"][0].cpu().numpy(), out["transmittance"][0].cpu().numpy()
    sigma, pcol = out["sigma"][0].cpu().numpy(), out["point_color"][0].cpu().numpy()
    valid = dbg["valid"].astype(bool)
    assert np.array_equal(sigma != 0, valid), "in-cube mask differs"
    np.testing.assert_allclose(sigma[valid], dbg["sigma"][valid], rtol=SIGMA_RTOL, atol=1e-6)
    e_s = float(np.max(np.abs(sigma[valid] - dbg["sigma"][valid]) / (np.abs(dbg["sigma"][valid]) + 1e-6)))
    e_p = np.abs(pcol[valid] - dbg["col"][valid]).max()
    assert e_p < PCOL_TOL
    e_t = np.abs(trans - o_trans).max()
    e_c = np.abs(color - o_color).max()
    e_r = np.abs(color - g["color"]).max()
    print(f"{name} split={int(split)}: max|T-oracle| {e_t:.2e}  max|color-oracle| {e_c:.2e}  max|color-reference| {e_r:.2e}  per-sample: sigma rel {e_s:.2e} colour {e_p:.2e}")
    assert e_t < T_TOL and e_c < PIX_TOL and e_r < PIX_TOL
    # deterministic
    out2 = m(torch.from_numpy(g["campos"])[None], torch.from_numpy(g["raydir"])[None], torch.from_numpy(g["bg"])[None],
             jitter_u=torch.from_numpy(g["U"])[None])
    assert torch.equal(out2["color"], out["color"])


def test_uv_no_background_and_short_chunks():
    from ngf_amd import uvmapping
    g, params = load_uv_case("uv_sphere")
    orc = OracleUV(params, "sphere")
    m = uvmapping.NeuTex(primitive_type="sphere", sample_num=24, device="cuda")
    m.load_params(params)
    U = g["U"][:7, :24].copy()
    o_color, o_trans = orc.render(g["campos"], g["raydir"][:7], U, bg=None)
    out = m(torch.from_numpy(g["campos"])[None], torch.from_numpy(g["raydir"][:7])[None], None, jitter_u=torch.from_numpy(U)[None])
    assert np.abs(out["color"][0].cpu().numpy() - o_color).max() < PIX_TOL
    assert np.abs(out["transmittance"][0].cpu().numpy() - o_trans).max() < T_TOL


def test_batch_of_cameras_is_one_launch_and_matches_per_camera_calls():
    """NeuTex.forward with N > 1 (model.py:27: camera_position [N,3], ray_direction [N,R,3], background_color [N,3]): ONE
    ngf_uv_render_batch launch with the cameras as a device table -- the same bits as N single-camera calls, as the host-pointer
    entry point ngf_uv_render, and the oracle's pixels per camera; an odd R makes a wave's two rays straddle two cameras."""
    import ctypes as C
    from ngf_amd import _lib, synth, uvmapping
    g, params = load_uv_case("uv_sphere")
    orc = OracleUV(params, "sphere")
    S, R = 32, 37
    net = uvmapping.NeuTex(primitive_type="sphere", sample_num=S, device="cuda")
    net.load_params(params)
    cams = np.stack([g["campos"], g["campos"] * np.float32(0.9) + np.float32(0.05), -g["campos"]]).astype(np.float32)
    bgs = np.stack([g["bg"], np.array([0.1, 0.7, 0.3], np.float32), np.zeros(3, np.float32)])
    dirs = np.stack([g["raydir"][:R], g["raydir"][R:2 * R], -g["raydir"][:R]]).astype(np.float32)
    U = synth.hash_uniform(12, 5, (3, R, S))
    out = net(torch.from_numpy(cams).cuda(), torch.from_numpy(dirs).cuda(), torch.from_numpy(bgs).cuda(), jitter_u=torch.from_numpy(U).cuda())
    L, h = _lib.lib(), net.handle()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in range(3):
        one = net(torch.from_numpy(cams[n:n + 1]), torch.from_numpy(dirs[n:n + 1]), torch.from_numpy(bgs[n:n + 1]), jitter_u=torch.from_numpy(U[n:n + 1]))
        assert torch.equal(one["color"][0], out["color"][n]) and torch.equal(one["transmittance"][0], out["transmittance"][n]), n
        # the host-pointer entry point (ngf_uv_render) on the same camera
        rd, Un = torch.from_numpy(dirs[n]).cuda(), torch.from_numpy(U[n]).cuda()
        col, tr = torch.empty((R, 3), device="cuda"), torch.empty((R,), device="cuda")
        _lib.check(L.ngf_uv_render(h, (C.c_float * 3)(*cams[n].tolist()), rd.data_ptr(), (C.c_float * 3)(*bgs[n].tolist()), Un.data_ptr(), R, S,
                                   col.data_ptr(), tr.data_ptr(), None, None, None, st))
        assert torch.equal(col, out["color"][n]) and torch.equal(tr, out["transmittance"][n]), n
        o_color, o_trans = orc.render(cams[n], dirs[n], U[n], bg=bgs[n])
        e_c = np.abs(out["color"][n].cpu().numpy() - o_color).max()
        e_t = np.abs(out["transmittance"][n].cpu().numpy() - o_trans).max()
        print(f"camera {n}: max|color-oracle| {e_c:.2e} max|T-oracle| {e_t:.2e}")
        # cameras 1, 2 are made-up poses (the conditioning of the path depends on the pose: 2.2e-5 measured for the mirrored camera):
        # north_star's bound for them, PIX_TOL for the reference's own camera
        assert e_c < (PIX_TOL if n == 0 else 1e-4) and e_t < T_TOL, (n, e_c, e_t)
    # no background: bg_dev NULL
    nb = net(torch.from_numpy(cams).cuda(), torch.from_numpy(dirs).cuda(), None, jitter_u=torch.from_numpy(U).cuda())
    o_color, _ = orc.render(cams[1], dirs[1], U[1], bg=None)
    assert np.abs(nb["color"][1].cpu().numpy() - o_color).max() < 1e-4


def test_uv_full_dtu_frame_properties():
    """BASELINE config 4 at full size: the whole 800 x 600 frame of the reference's DTU camera 0 (480 000 rays generated on the
    device by ngf_generate_rays_dtu, 64 samples per ray, sphere gauge, seeded weights -- no checkpoint is obtainable offline).
    Finite, in range, any subset rendered alone gives the same bits, and a strided sample of the frame against the C oracle."""
    from ngf_amd import rays as nrays
    from ngf_amd import synth, uvmapping
    params = synth.uvmapping_params(5, "sphere")
    v0 = synth.DTU_VIEW0
    dirs = nrays.generate_rays_dtu(600, 800, v0["focal"], v0["princpt"], v0["rot"])          # [480000, 3]
    n = dirs.shape[0]
    assert n == 480000
    cam = torch.tensor(v0["campos"], dtype=torch.float32)[None]
    bg = torch.tensor([[0.2, 0.5, 0.9]], dtype=torch.float32)
    gen = torch.Generator(device="cuda").manual_seed(11)
    U = torch.rand((1, n, 64), device="cuda", generator=gen)
    net = uvmapping.NeuTex(primitive_type="sphere", sample_num=64, device="cuda")
    net.load_params(params)
    out = net(cam, dirs[None], bg, jitter_u=U, collect_stats=True)
    color, trans = out["color"][0], out["transmittance"][0]
    st = net.last_stats.cpu().numpy()
    assert 0 < st[0] <= n * 64                                   # in-cube samples evaluated
    assert bool(torch.isfinite(color).all()) and float(color.min()) >= 0.0 and float(color.max()) <= 1.0
    assert float(trans.min()) >= 0.0 and float(trans.max()) <= 1.0 + 1e-6
    idx = torch.arange(0, n, 997, device="cuda")
    sub = net(cam, dirs[idx][None], bg, jitter_u=U[:, idx])
    assert torch.equal(sub["color"][0], color[idx]) and torch.equal(sub["transmittance"][0], trans[idx])
    pick = idx[::2].cpu().numpy()                                # ~240 rays through the whole frame
    orc = OracleUV(params, "sphere")
    o_color, o_trans = orc.render(np.asarray(v0["campos"], np.float32), dirs[pick].cpu().numpy(), U[0, pick].cpu().numpy(), bg=bg[0].numpy())
    e_c = np.abs(color[pick].cpu().numpy() - o_color).max()
    e_t = np.abs(trans[pick].cpu().numpy() - o_trans).max()
    print(f"full DTU frame: {st[0] / n:.1f} in-cube samples per ray, max|color-oracle| {e_c:.2e}, max|T-oracle| {e_t:.2e} on {len(pick)} rays")
    assert e_c < 5e-5 and e_t < T_TOL                            # north_star: 1e-4


def test_two_rays_per_wave_is_bit_identical():
    """The default kernel renders two rays per wave (every weight load feeds two MFMAs); a sample's arithmetic does not
    depend on its tile, so the one-ray-per-wave kernel (knob uv_tiles = 1) gives the same bits -- also for an odd ray count
    (the last wave's second ray is a dummy) and a single ray."""
    import ngf_amd  # noqa: F401
    from ngf_amd import _lib, synth, uvmapping
    net = uvmapping.NeuTex(primitive_type="sphere", sample_num=64)
    net.load_params(synth.uvmapping_params(31, "sphere"))
    campos, dirs = synth.dtu_rays(600, 800)
    for n in (1, 37, 64):
        pick = (synth.hash_uniform(8, n, (n,)) * np.float32(dirs.shape[0])).astype(np.int64)
        rd = torch.from_numpy(dirs[pick])[None].cuda()
        cp = torch.from_numpy(campos)[None].cuda()
        U = torch.from_numpy(synth.hash_uniform(8, 100 + n, (1, n, 64))).cuda()
        two = net(cp, rd, None, jitter_u=U)
        with _lib.knobs(uv_tiles=1):
            one = net(cp, rd, None, jitter_u=U)
        assert torch.equal(two["color"], one["color"]) and torch.equal(two["transmittance"], one["transmittance"]), n


@pytest.mark.parametrize("name", ["uv_sphere", "uv_square"])
def test_uv_split_bf16_keeps_the_fp32_tolerances(name):
    """NGF_UV_F_SPLIT_BF16 (opt-in): the eighteen 256 -> 256 layers and block2.0 as six bf16 MFMA products per fp32 product
    (3-term split operands, fp32 accumulate; csrc/ngf_uv.hpp dense_bf16).  The path is ill-conditioned by construction, so the
    bar is the one the fp32-MFMA kernel is held to: the same tolerances against the oracle and the reference's golden pixels --
    and the split kernel must sit as close to the oracle as the fp32 kernel does (its rounding noise is of the same size)."""
    from ngf_amd import uvmapping
    g, params = load_uv_case(name)
    pt = str(g["primitive_type"])
    orc = OracleUV(params, pt)
    o_color, o_trans, dbg = orc.render(g["campos"], g["raydir"], g["U"], bg=g["bg"], debug=True)
    args = (torch.from_numpy(g["campos"])[None], torch.from_numpy(g["raydir"])[None], torch.from_numpy(g["bg"])[None])
    res = {}
    for split in (False, True):
        m = uvmapping.NeuTex(primitive_type=pt, sample_num=int(g["S"]), device="cuda", split_bf16=split)
        m.load_params(params)
        out = m(*args, jitter_u=torch.from_numpy(g["U"])[None], debug=True)
        res[split] = (out["color"][0].cpu().numpy(), out["transmittance"][0].cpu().numpy(), out["sigma"][0].cpu().numpy(),
                      out["point_color"][0].cpu().numpy())
        m.release()
    valid = dbg["valid"].astype(bool)
    color, trans, sigma, pcol = res[True]
    assert np.array_equal(sigma != 0, valid), "in-cube mask differs"
    np.testing.assert_allclose(sigma[valid], dbg["sigma"][valid], rtol=SIGMA_RTOL, atol=1e-6)
    assert np.abs(pcol[valid] - dbg["col"][valid]).max() < PCOL_TOL
    e = {s: (np.abs(res[s][1] - o_trans).max(), np.abs(res[s][0] - o_color).max(), np.abs(res[s][0] - g["color"]).max()) for s in (False, True)}
    print(f"{name} split={int(split)}: max|T-oracle| fp32 {e[False][0]:.2e} split {e[True][0]:.2e}; max|color-oracle| fp32 {e[False][1]:.2e} split {e[True][1]:.2e}; "
          f"max|color-reference| fp32 {e[False][2]:.2e} split {e[True][2]:.2e}; max|color split - fp32| {np.abs(res[True][0] - res[False][0]).max():.2e}")
    assert e[True][0] < T_TOL and e[True][1] < PIX_TOL and e[True][2] < PIX_TOL
    assert e[True][1] < 3 * e[False][1] + 2e-5             # no worse than the fp32 kernel's own rounding noise (x3 slack)
    with pytest.raises(RuntimeError):
        from ngf_amd import _lib
        m = uvmapping.NeuTex(primitive_type=pt, sample_num=int(g["S"]), device="cuda", split_bf16=True)
        m.load_params(params)
        with _lib.knobs(uv_tiles=1):
            m(*args, jitter_u=torch.from_numpy(g["U"])[None])
