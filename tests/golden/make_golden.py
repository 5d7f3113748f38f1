#!/usr/bin/env python3
"""Capture golden vectors FROM THE REFERENCE ITSELF (run in the build container only).

    python tests/golden/make_golden.py          # writes tests/golden/*.npz

Imports /root/reference/{TriPlane,InfoInv}/models read-only (torch CPU), loads seeded
parameters from ``ngf_amd.synth`` into the reference modules by attribute assignment (the same
way the reference's own ``up_sampling`` replaces planes, TriPlane/models/Field.py:110-112), runs
the reference ``forward`` and a few of its sub-functions, and stores inputs + outputs.

The fixtures hold DATA only: rays, scalar configuration, parameter checksums (the parameters
themselves are regenerated bit-exactly from the seed by ``synth``) and the reference's outputs.
/root/reference never travels to the GPU box; these files do.
"""
import contextlib
import importlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import ngf_amd  # noqa: E402
from ngf_amd import synth  # noqa: E402

REF = "/root/reference"


def _import_ref(subdir):
    """Import <subdir>/models.Field from the reference with a clean ``models`` namespace."""
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(REF, subdir))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mod = importlib.import_module("models.Field")
    finally:
        sys.path.pop(0)
    return mod


def _checksums(params):
    out = {}
    for k, v in params.items():
        v64 = v.astype(np.float64).reshape(-1)
        out["chk." + k] = np.array([v64.sum(), np.abs(v64).sum(), v64[:: max(1, v64.size // 7)][:7].sum()])
    return out


def _load_params(field, params):
    for k, v in params.items():
        obj = field
        parts = k.split(".")
        for p in parts[:-1]:
            obj = getattr(obj, p) if not p.isdigit() else obj[int(p)]
        setattr(obj, parts[-1], torch.nn.Parameter(torch.from_numpy(v.copy())))


def _rays_for_case(seed, n_frame=160, n_edge=64):
    frame = synth.lookat_rays(40, 40)
    pick = (synth.hash_uniform(seed, 400, (n_frame,)) * np.float32(frame.shape[0])).astype(np.int64)
    return np.concatenate([frame[pick], synth.edge_rays(seed, n_edge)], 0)


def capture_triplane(name, seed, preset, gauge_on, gauge_std, with_mask, S, white_bg=True, train=None):
    """train = (coin,): forward(is_train=True) with the per-ray jitter of sample_ray (FieldBase.py:128-130) supplied through a patched
    torch.rand_like and the background coin of FieldBase.py:299 through a patched torch.rand."""
    F = _import_ref("TriPlane")
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    grid = [24, 20, 18]
    plane_hw = ((20, 24), (18, 20), (18, 24))   # plane_xy [Ny,Nx], plane_yz [Nz,Ny], plane_xz [Nz,Nx]
    gauge_hw = (12, 14)
    params = synth.triplane_params(seed, plane_hw, gauge_hw, preset=preset, gauge_std=gauge_std)
    with contextlib.redirect_stdout(io.StringIO()):
        field = F.TriPlane(aabb, grid, "cpu", near_far=[2.0, 6.0], alphaMask_thres=1e-4, distance_scale=25,
                           rayMarch_weight_thres=1e-4, step_ratio=0.5, gauge_start=0)
    _load_params(field, params)
    extra = {}
    if with_mask:
        dhw = (10, 12, 14)
        vol, bits = synth.alpha_mask_bits(seed, dhw)
        maabb = torch.tensor([[-1.4, -1.3, -1.45], [1.35, 1.5, 1.2]])
        field.alphaMask = F.AlphaGridMask("cpu", maabb, torch.from_numpy(vol.astype(np.float32)))
        extra = {"mask_bits": bits, "mask_dhw": np.array(dhw), "mask_aabb": maabb.numpy()}
    rays = _rays_for_case(seed)
    iteration = 30001 if gauge_on else -1
    field.gauge_start = 0
    is_train = train is not None
    U = synth.hash_uniform(seed, 820, (rays.shape[0], 1))
    real_like, real_rand = torch.rand_like, torch.rand
    if is_train:
        extra.update({"jitter": U[:, 0], "coin": np.float32(train[0]), "is_train": np.array(1)})
    with torch.no_grad():
        if is_train:
            torch.rand_like = lambda *a, **k: torch.from_numpy(U.copy())
            torch.rand = lambda *a, **k: torch.tensor([float(train[0])])
        try:
            out = field(torch.from_numpy(rays), white_bg=white_bg, is_train=is_train, N_samples=S, iteration=iteration)
        finally:
            torch.rand_like, torch.rand = real_like, real_rand
        # intermediates of the first 8 rays through the reference's own sub-functions
        r8 = torch.from_numpy(rays[:8])
        if is_train:
            torch.rand_like = lambda *a, **k: torch.from_numpy(U[:8].copy())
        try:
            pts, z, valid = field.sample_ray(r8[:, :3], r8[:, 3:6], is_train=is_train, N_samples=S)
        finally:
            torch.rand_like = real_like
        if with_mask:
            a = field.alphaMask.sample_alpha(pts[valid])
            inv = ~valid
            inv[valid] |= ~(a > 0)
            valid = ~inv
        xyzn = field.normalize_coord(pts)
        sigma = torch.zeros(pts.shape[:-1])
        coords = torch.zeros((*pts.shape[:2], 6))
        if valid.any():
            txy, tyz, txz = field.compute_gauge(xyzn[valid], iteration=iteration)
            sigma[valid] = field.compute_density(txy, tyz, txz)
            coords[valid] = torch.cat([txy, tyz, txz], -1)
        dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), -1)
        alpha, weight, _ = F.raw2alpha(sigma, dists * field.distance_scale)
        # per-sample colours of the active samples (FieldBase.py:289-294): rgb_mask and compute_rgb -> rgb_decoder, the reference's own calls
        rgb_mask = weight > field.rayMarch_weight_thres
        rgb = torch.zeros((*pts.shape[:2], 3))
        vd = r8[:, 3:6].view(-1, 1, 3).expand(pts.shape)
        if rgb_mask.any():
            rgb[rgb_mask] = field.compute_rgb(coords[..., 0:2][rgb_mask], coords[..., 2:4][rgb_mask], coords[..., 4:6][rgb_mask], vd[rgb_mask])
        extra.update({"i_rgb_mask": rgb_mask.numpy(), "i_rgb": rgb.numpy()})
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), model="triplane", seed=seed, preset=preset, gauge_on=int(gauge_on),
        gauge_std=np.float32(gauge_std), S=S, white_bg=int(white_bg), aabb=aabb.numpy(), grid=np.array(grid),
        plane_hw=np.array(plane_hw), gauge_hw=np.array(gauge_hw), near_far=np.array([2.0, 6.0], np.float32),
        step_ratio=np.float32(0.5), distance_scale=np.float32(25), thr=np.float32(1e-4),
        stepSize=field.stepSize.numpy(), nSamples=field.nSamples, rays=rays,
        rgb_map=out["rgb_map"].numpy(), depth_map=out["depth_map"].numpy(),
        i_z=z.numpy(), i_valid=valid.numpy(), i_sigma=sigma.numpy(), i_coords=coords.numpy(),
        i_alpha=alpha.numpy(), i_weight=weight.numpy(), **_checksums(params), **extra)
    act = float((weight > 1e-4).float().mean())
    print(f"{name}: rays {rays.shape[0]} S {S} mean rgb {out['rgb_map'].mean():.5f} active(first 8) {act:.3f}")


def capture_infoinv(name, seed, preset, infoinv, S, with_mask=False, white_bg=True):
    F = _import_ref("InfoInv")
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    grid = [22, 22, 22]
    plane_hw = ((16, 18), (14, 16), (14, 18))
    params = synth.infoinv_params(seed, plane_hw, preset=preset)
    with contextlib.redirect_stdout(io.StringIO()):
        field = F.TriPlane(aabb, grid, "cpu", near_far=[2.0, 6.0], alphaMask_thres=1e-4, distance_scale=25,
                           rayMarch_weight_thres=1e-4, step_ratio=0.5)
    _load_params(field, params)
    extra = {}
    if with_mask:      # the alpha-mask branch of the InfoInv forward (InfoInv/models/FieldBase.py:238-244)
        dhw = (9, 11, 13)
        vol, bits = synth.alpha_mask_bits(seed, dhw)
        maabb = torch.tensor([[-1.45, -1.35, -1.4], [1.3, 1.5, 1.25]])
        field.alphaMask = F.AlphaGridMask("cpu", maabb, torch.from_numpy(vol.astype(np.float32)))
        extra = {"mask_bits": bits, "mask_dhw": np.array(dhw), "mask_aabb": maabb.numpy()}
    rays = _rays_for_case(seed, 96, 32)
    with torch.no_grad():
        out = field(torch.from_numpy(rays), white_bg=white_bg, is_train=False, N_samples=S, infoinv=infoinv)
        r8 = torch.from_numpy(rays[:8])
        pts, z, valid = field.sample_ray(r8[:, :3], r8[:, 3:6], is_train=False, N_samples=S)
        if with_mask:
            a = field.alphaMask.sample_alpha(pts[valid])
            inv = ~valid
            inv[valid] |= ~(a > 0)
            valid = ~inv
        xyzn = field.normalize_coord(pts)
        sigma = torch.zeros(pts.shape[:-1])
        coords = torch.zeros((*pts.shape[:2], 6))
        if valid.any():
            txy, tyz, txz = field.transform(xyzn[valid])
            sigma[valid] = field.compute_density(txy, tyz, txz, infoinv=infoinv)
            coords[valid] = torch.cat([txy, tyz, txz], -1)
        dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), -1)
        alpha, weight, _ = F.raw2alpha(sigma, dists * field.distance_scale)
        # per-sample colours of the active samples (InfoInv/models/FieldBase.py:262-266): the reference's own compute_rgb
        rgb_mask = weight > field.rayMarch_weight_thres
        rgb = torch.zeros((*pts.shape[:2], 3))
        vd = r8[:, 3:6].view(-1, 1, 3).expand(pts.shape)
        if rgb_mask.any():
            rgb[rgb_mask] = field.compute_rgb(coords[..., 0:2][rgb_mask], coords[..., 2:4][rgb_mask], coords[..., 4:6][rgb_mask], vd[rgb_mask],
                                              infoinv=infoinv)
        extra.update({"i_rgb_mask": rgb_mask.numpy(), "i_rgb": rgb.numpy(), "i_coords": coords.numpy()})
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), model="infoinv", seed=seed, preset=preset, infoinv=int(infoinv), S=S,
        white_bg=int(white_bg), aabb=aabb.numpy(), grid=np.array(grid), plane_hw=np.array(plane_hw),
        near_far=np.array([2.0, 6.0], np.float32), step_ratio=np.float32(0.5), distance_scale=np.float32(25),
        thr=np.float32(1e-4), stepSize=field.stepSize.numpy(), nSamples=field.nSamples, rays=rays,
        rgb_map=out["rgb_map"].numpy(), depth_map=out["depth_map"].numpy(),
        i_z=z.numpy(), i_valid=valid.numpy(), i_sigma=sigma.numpy(), i_alpha=alpha.numpy(),
        i_weight=weight.numpy(), **_checksums(params), **extra)
    print(f"{name}: rays {rays.shape[0]} S {S} mean rgb {out['rgb_map'].mean():.5f} active(first 8) {float(rgb_mask.float().mean()):.3f}")


def capture_ops(name="ops_grid_sample"):
    """Pin the two ATen samplers on their own: rectangular plane, out-of-range coordinates."""
    import torch.nn.functional as Fn
    plane = synth.hash_normal(5, 1, (1, 5, 7, 9))
    uv = (synth.hash_uniform(5, 2, (300, 2)) * np.float32(2.6) - np.float32(1.3))
    uv[:8] = np.array([[-1, -1], [1, 1], [1, -1], [-1, 1], [0, 0], [1.0000001, 0], [-1.0000001, 0.5], [0.999999, 1]])
    out2 = Fn.grid_sample(torch.from_numpy(plane), torch.from_numpy(uv).view(1, -1, 1, 2), align_corners=True)
    out2 = out2.view(5, -1).T.contiguous().numpy()
    vol, bits = synth.alpha_mask_bits(6, (5, 6, 7))
    q = (synth.hash_uniform(5, 3, (400, 3)) * np.float32(2.4) - np.float32(1.2))
    q[:4] = np.array([[-1, -1, -1], [1, 1, 1], [0, 0, 0], [1, -1, 0.5]])
    out3 = Fn.grid_sample(torch.from_numpy(vol.astype(np.float32)).view(1, 1, 5, 6, 7),
                          torch.from_numpy(q).view(1, -1, 1, 1, 3), align_corners=True).view(-1).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), plane=plane, uv=uv, out2=out2,
                        mask_bits=bits, mask_dhw=np.array([5, 6, 7]), q=q, out3=out3)
    print(f"{name}: 2-D {out2.shape}, 3-D {out3.shape} (positive {int((out3 > 0).sum())})")


def capture_alpha_mask(name="triplane_alpha_mask", seed=41):
    """updateAlphaMask + filtering_rays of the reference on a small lattice (FieldBase.py:161-246)."""
    F = _import_ref("TriPlane")
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    grid = [24, 20, 18]
    plane_hw = ((20, 24), (18, 20), (18, 24))
    gauge_hw = (12, 14)
    params = synth.triplane_params(seed, plane_hw, gauge_hw, preset="R2", gauge_std=0.05)
    with contextlib.redirect_stdout(io.StringIO()):
        field = F.TriPlane(aabb, grid, "cpu", near_far=[2.0, 6.0], alphaMask_thres=0.02, distance_scale=25,
                           rayMarch_weight_thres=1e-4, step_ratio=0.5, gauge_start=0)
    _load_params(field, params)
    mgrid = (14, 12, 10)
    rays = _rays_for_case(seed, 200, 64)
    rgbs = synth.hash_uniform(seed, 700, (rays.shape[0], 3))
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        dense_alpha, _ = field.getDenseAlpha(mgrid)
        new_aabb = field.updateAlphaMask(mgrid)
        vol = field.alphaMask.alpha_volume[0, 0].numpy().copy()
        kept_rays, kept_rgbs = field.filtering_rays(torch.from_numpy(rays), torch.from_numpy(rgbs), N_samples=40)
        kept_bbox, _ = field.filtering_rays(torch.from_numpy(rays), torch.from_numpy(rgbs), bbox_only=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), model="triplane", seed=seed, preset="R2", gauge_std=np.float32(0.05),
                        aabb=aabb.numpy(), grid=np.array(grid), plane_hw=np.array(plane_hw), gauge_hw=np.array(gauge_hw),
                        near_far=np.array([2.0, 6.0], np.float32), step_ratio=np.float32(0.5), distance_scale=np.float32(25),
                        thr=np.float32(1e-4), alphaMask_thres=np.float32(0.02), mgrid=np.array(mgrid), rays=rays, rgbs=rgbs,
                        dense_alpha=dense_alpha.numpy(), mask_volume=vol, new_aabb=new_aabb.numpy(),
                        kept_rays=kept_rays.numpy(), kept_bbox=kept_bbox.numpy(), **_checksums(params))
    print(f"{name}: occupancy {vol.mean():.3f}, new aabb {new_aabb.numpy().round(3).tolist()}, kept {kept_rays.shape[0]}/{rays.shape[0]} "
          f"(bbox only {kept_bbox.shape[0]})")


def capture_infoinv_alpha(name="infoinv_alpha_mask", seed=43):
    """InfoInv tree: compute_alpha / getDenseAlpha / updateAlphaMask with infoinv=True and infoinv=False
    (InfoInv/models/FieldBase.py:140-193; main.py:325 passes the flag through) on a small lattice."""
    F = _import_ref("InfoInv")
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    grid = [22, 22, 22]
    plane_hw = ((16, 18), (14, 16), (14, 18))
    params = synth.infoinv_params(seed, plane_hw, preset="R1")
    mgrid = (12, 11, 10)
    pts = (synth.hash_uniform(seed, 710, (150, 3)) * np.float32(3.4) - np.float32(1.7)).astype(np.float32)
    out = {}
    for flag in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            field = F.TriPlane(aabb, grid, "cpu", near_far=[2.0, 6.0], alphaMask_thres=0.06, distance_scale=25,
                               rayMarch_weight_thres=1e-4, step_ratio=0.5)
        _load_params(field, params)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            a_pts = field.compute_alpha(torch.from_numpy(pts), 0.37, infoinv=flag)
            dense_alpha, _ = field.getDenseAlpha(mgrid, infoinv=flag)
            field.updateAlphaMask(mgrid, infoinv=flag)
        tag = "on" if flag else "off"
        out.update({f"alpha_pts_{tag}": a_pts.numpy(), f"dense_alpha_{tag}": dense_alpha.numpy(),
                    f"mask_volume_{tag}": field.alphaMask.alpha_volume[0, 0].numpy().copy()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), model="infoinv", seed=seed, preset="R1", aabb=aabb.numpy(), grid=np.array(grid),
                        plane_hw=np.array(plane_hw), near_far=np.array([2.0, 6.0], np.float32), step_ratio=np.float32(0.5),
                        distance_scale=np.float32(25), thr=np.float32(1e-4), alphaMask_thres=np.float32(0.06), mgrid=np.array(mgrid), pts=pts,
                        **out, **_checksums(params))
    print(f"{name}: occupancy on {out['mask_volume_on'].mean():.3f} off {out['mask_volume_off'].mean():.3f}, "
          f"max |alpha on - off| {np.abs(out['dense_alpha_on'] - out['dense_alpha_off']).max():.3f}")


def capture_uv(name, seed, primitive_type, R=96, S=64):
    """UV-Mapping colour path through the reference's own sub-modules, composed as NeuTex.forward does
    (model.py:30-50); NeuTex.forward itself is CUDA-hardwired (gauge_fields.py:129,154) and cannot run here.
    torch.rand inside cube_ray_generation is replaced by a stored uniform tensor so the jitter is reproducible."""
    for k in [k for k in sys.modules if k in ("model", "util") or k.startswith("model.")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(REF, "UV-Mapping"))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            dec = importlib.import_module("model.decoder")
            gf = importlib.import_module("model.gauge_fields")
            rn = importlib.import_module("model.renderer")
    finally:
        sys.path.pop(0)
    params = synth.uvmapping_params(seed, primitive_type)
    with contextlib.redirect_stdout(io.StringIO()):
        geo = dec.GeometryMlpDecoder(pos_freqs=10, hidden_size=256, num_layers=10)
        gauge = gf.GaugeTransform(primitive_type)
        tex = dec.TextureMlpDecoder(3, 10, 6, uv_dim=2 if primitive_type == "square" else 3, layers=[5, 3], width=256,
                                    clamp=False, primitive_type=primitive_type, target_texture="None")
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    geo.load_state_dict({k[len("net_geometry_decoder."):]: v for k, v in sd.items() if k.startswith("net_geometry_decoder.")})
    gauge.load_state_dict({k[len("gauge_transform."):]: v for k, v in sd.items() if k.startswith("gauge_transform.")})
    tex.load_state_dict({k[len("net_texture."):]: v for k, v in sd.items() if k.startswith("net_texture.")})
    campos, dirs = synth.dtu_rays(600, 800)
    pick = (synth.hash_uniform(seed, 600, (R,)) * np.float32(dirs.shape[0])).astype(np.int64)
    raydir = dirs[pick]
    raydir[-4:] = np.array([[0.3, 0.9, 0.3], [-0.6, 0.1, 0.79], [0, 0, 1], [0.577, 0.577, 0.578]], np.float32)  # misses etc.
    raydir = raydir / np.linalg.norm(raydir, axis=1, keepdims=True)
    U = synth.hash_uniform(seed, 601, (1, R, S))
    bg = np.array([0.2, 0.5, 0.8], np.float32)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: torch.from_numpy(U.copy())
    try:
        with torch.no_grad():
            cp, rd = torch.from_numpy(campos)[None], torch.from_numpy(raydir)[None]
            ray_pos, ray_dist, ray_valid, _ = rn.cube_ray_generation(cp, rd, S, jitter=0.05)
    finally:
        torch.rand = real_rand
    with torch.no_grad():
        density = geo(ray_pos)["density"][..., None]
        uv = gauge(ray_pos)
        feats = tex(uv, rd[:, :, None, :])
        bsdf = torch.cat([density, feats[..., :3]], -1)
        out = rn.ray_march(rd, ray_pos, ray_dist, ray_valid, bsdf, None, None, rn.radiance_render, rn.alpha_blend)
        ray_color, bgw = out[0], out[6]
        ray_color = ray_color + torch.from_numpy(bg)[None, None, :] * bgw[:, :, None]
        color = rn.simple_tone_map(ray_color)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), model="uv", seed=seed, primitive_type=primitive_type, S=S,
                        campos=campos, raydir=raydir, U=U[0], bg=bg, color=color[0].numpy(), transmittance=bgw[0].numpy(),
                        i_sigma=density[0, :8, :, 0].numpy(), i_uv=uv[0, :8].numpy(), i_col=feats[0, :8, :, :3].numpy(),
                        i_valid=ray_valid[0, :8].numpy(), **_checksums(params))
    print(f"{name}: rays {R} S {S} mean color {float(color.mean()):.4f} mean T {float(bgw.mean()):.4f} valid {float(ray_valid.float().mean()):.3f}")


def capture_evalout():
    """Eval output stage: the reference's own rgb_ssim and visualize_depth_numpy (TriPlane/utils.py) run on
    seeded inputs.  utils.py does not import here (cv2, imageio, ... are absent), so the two function definitions are
    taken out of the module's syntax tree and executed with numpy/scipy; cv2.applyColorMap is the identity, i.e. the
    captured depth image is the uint8 INDEX image the colour table is applied to (the table itself cannot be pinned)."""
    import ast
    import types
    import scipy.signal
    src = open(os.path.join(REF, "TriPlane", "utils.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("rgb_ssim", "visualize_depth_numpy")]
    assert len(keep) == 2
    cv2 = types.SimpleNamespace(COLORMAP_JET=2, applyColorMap=lambda x, cmap: x)
    import scipy
    ns = {"np": np, "scipy": scipy, "cv2": cv2, "torch": torch}
    # default argument cv2.COLORMAP_JET is evaluated at definition time -> cv2 must be in the namespace first
    exec(compile(ast.Module(body=keep, type_ignores=[]), "reference:TriPlane/utils.py", "exec"), ns)
    H, W = 37, 45
    img0 = synth.hash_uniform(41, 700, (H, W, 3))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    smooth = (0.5 + 0.5 * np.sin(xx / 5.0)[..., None] * np.cos(yy / 7.0)[..., None]).astype(np.float32)
    img0 = (0.7 * smooth + 0.3 * img0).astype(np.float32)
    img1 = np.clip(img0 + (synth.hash_uniform(41, 701, (H, W, 3)) - np.float32(0.5)) * np.float32(0.2), 0, 1).astype(np.float32)
    img1[:6, :8] = img0[:6, :8]                       # a patch with zero error
    img1[-5:, -5:] = 0.25                             # constant patch: zero variance in img1
    t0, t1 = torch.from_numpy(img0), torch.from_numpy(img1)      # the reference passes torch CPU tensors (main.py:109)
    ssim = ns["rgb_ssim"](t0, t1, 1)
    ssim_map = ns["rgb_ssim"](t0, t1, 1, return_map=True)
    ssim5 = ns["rgb_ssim"](t0, t1, 1, filter_size=5, filter_sigma=0.8)
    loss = torch.mean((t0 - t1) ** 2)
    psnr = -10.0 * np.log(loss.item()) / np.log(10.0)
    depth = (np.float32(2.0) + np.float32(4.5) * synth.hash_uniform(41, 702, (24, 20))).astype(np.float32)
    depth[0, :6] = [0.0, -0.4, 1.2, 6.7, np.nan, np.inf]       # background / below near / above far / non-finite
    depth[1, 0] = -np.inf
    d_nf, mm_nf = ns["visualize_depth_numpy"](depth.copy(), (2.0, 6.0))
    finite = depth.copy()
    finite[~np.isfinite(finite)] = 3.0
    d_auto, mm_auto = ns["visualize_depth_numpy"](finite.copy(), None)
    rgb = (synth.hash_uniform(41, 703, (16, 12, 3)) * np.float32(1.4) - np.float32(0.2)).astype(np.float32)
    rgb8 = (torch.from_numpy(rgb).clamp(0.0, 1.0).numpy() * 255).astype('uint8')          # main.py:98,117
    np.savez_compressed(os.path.join(HERE, "evalout.npz"), img0=img0, img1=img1, ssim=np.float64(ssim), ssim_map=ssim_map,
                        ssim5=np.float64(ssim5), mse=np.float64(loss.item()), psnr=np.float64(psnr), depth=depth, depth_idx_nearfar=d_nf,
                        depth_finite=finite, depth_idx_auto=d_auto, depth_auto_range=np.asarray(mm_auto, np.float64), rgb=rgb, rgb8=rgb8)
    print(f"evalout: ssim {ssim:.6f} ssim5 {ssim5:.6f} psnr {psnr:.4f} depth idx mean {d_nf.mean():.2f} auto range {mm_auto}")


def _ref_functions(relpath, names, ns):
    """Execute the named top-level function definitions of a reference source file (which does not import here) from
    its syntax tree, in namespace ``ns``.  Nothing of the reference's text is stored: only the functions' outputs."""
    import ast
    tree = ast.parse(open(os.path.join(REF, relpath)).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), (relpath, names)
    exec(compile(ast.Module(body=keep, type_ignores=[]), "reference:" + relpath, "exec"), ns)
    return ns


def capture_rays():
    """Ray generation FROM THE REFERENCE's own functions.

    rays_blender.npz: get_ray_directions + get_rays (TriPlane/dataLoader/ray_utils.py:24-42,66-87) executed from the
    module's syntax tree (the module imports kornia, absent here), with the Blender loader's call pattern
    (blender.py:46-53,84-85): focal from camera_angle_x, directions / torch.norm(directions, dim=-1, keepdim=True),
    rays = cat(rays_o, rays_d).  kornia.create_meshgrid is supplied the way kornia defines it:
    create_meshgrid(H, W, normalized_coordinates=False) = [1,H,W,2] with [...,0] = x = linspace(0, W-1, W) along
    columns and [...,1] = y = linspace(0, H-1, H) along rows.
    rays_dtu.npz: get_rays_dir (UV-Mapping/data/dtu.py:27-37) on the 'no_crop' pixel grid of DtuDataset.__getitem__
    (dtu.py:160-169: integer pixel coordinates, float32) with the cameras the reference ships
    (UV-Mapping/data/DTU/scan83/trainData/in_cam*.npy, views 0 and 33).
    Fixtures keep whole image rows (what ngf_generate_rays* produces) of a few rows only."""
    def create_meshgrid(H, W, normalized_coordinates=False):
        assert not normalized_coordinates
        xs = torch.linspace(0, W - 1, W)
        ys = torch.linspace(0, H - 1, H)
        return torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1).permute(1, 0, 2).unsqueeze(0)

    ns = _ref_functions("TriPlane/dataLoader/ray_utils.py", ("get_ray_directions", "get_rays"), {"torch": torch, "np": np, "create_meshgrid": create_meshgrid})
    out = {}
    cases = (("a", 800, 800, 0.6911112070083618, synth.lookat_pose(), (0, 1, 2, 399, 400, 401, 798, 799)),
             ("b", 37, 53, 0.9, synth.lookat_pose(3.2, -20.0, 250.0), tuple(range(37))))
    for tag, H, W, angle, c2w, rows in cases:
        focal = 0.5 * 800 / np.tan(0.5 * angle)            # blender.py:46 (np.float64 scalar, like the loader's)
        focal *= W / 800                                   # blender.py:47
        directions = ns["get_ray_directions"](H, W, [focal, focal])                    # blender.py:51
        directions = directions / torch.norm(directions, dim=-1, keepdim=True)         # blender.py:52
        rays_o, rays_d = ns["get_rays"](directions, torch.FloatTensor(c2w))            # blender.py:68,84
        rays = torch.cat([rays_o, rays_d], 1).reshape(H, W, 6).numpy()                 # blender.py:85
        out.update({f"{tag}_H": H, f"{tag}_W": W, f"{tag}_focal": np.float64(focal), f"{tag}_c2w": np.asarray(c2w, np.float32),
                    f"{tag}_rows": np.array(rows), f"{tag}_rays": np.ascontiguousarray(rays[list(rows)])})
    np.savez_compressed(os.path.join(HERE, "rays_blender.npz"), **out)
    print("rays_blender:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith("rays")})

    ns = _ref_functions("UV-Mapping/data/dtu.py", ("get_rays_dir",), {"np": np})
    cam = os.path.join(REF, "UV-Mapping", "data", "DTU", "scan83", "trainData")
    focal_all, princpt_all = np.load(os.path.join(cam, "in_camFocal.npy")), np.load(os.path.join(cam, "in_camPrincpt.npy"))
    ext_all, org_all = np.load(os.path.join(cam, "in_camExtrinsics.npy")), np.load(os.path.join(cam, "in_camOrgs.npy"))
    H, W = 600, 800
    out = {"H": H, "W": W}
    for view, rows in ((0, (0, 1, 299, 300, 598, 599)), (33, (100, 101, 455))):
        px, py = np.meshgrid(np.arange(W).astype(np.float32), np.arange(H).astype(np.float32))     # dtu.py:160-163
        pixelcoords = np.stack((px, py), axis=-1).astype(np.float32)                                 # dtu.py:165
        camrot = ext_all[view][0:3, 0:3]                                                             # dtu.py:133
        raydir = ns["get_rays_dir"](pixelcoords, H, W, focal_all[view], camrot, princpt_all[view])   # dtu.py:166-168
        assert raydir.dtype == np.float32
        out.update({f"v{view}_focal": focal_all[view], f"v{view}_princpt": princpt_all[view], f"v{view}_rot": camrot,
                    f"v{view}_campos": org_all[view].astype(np.float32),                             # dtu.py:136 (.float())
                    f"v{view}_rows": np.array(rows), f"v{view}_raydir": np.ascontiguousarray(raydir[list(rows)])})
    np.savez_compressed(os.path.join(HERE, "rays_dtu.npz"), **out)
    print("rays_dtu:", {k: v.shape for k, v in out.items() if k.endswith("raydir")})


def capture_train(name="train_r1", seed=51, S=40, steps=2):
    """Training step: the reference TriPlane module in training mode -- forward(is_train=True) with the
    per-ray jitter supplied through a patched torch.rand_like, the loss of main.py:281-291, autograd, and ``steps``
    iterations of torch.optim.Adam over get_optparam_groups with the lr decay of main.py:298-299."""
    F = _import_ref("TriPlane")
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    grid = [12, 10, 9]
    plane_hw = ((10, 12), (9, 10), (9, 12))
    gauge_hw = (7, 8)
    params = synth.triplane_params(seed, plane_hw, gauge_hw, preset="R1", gauge_std=0.05)
    with contextlib.redirect_stdout(io.StringIO()):
        field = F.TriPlane(aabb, grid, "cpu", near_far=[2.0, 6.0], alphaMask_thres=1e-4, distance_scale=25,
                           rayMarch_weight_thres=1e-4, step_ratio=0.5, gauge_start=0)
    _load_params(field, params)
    rays = _rays_for_case(seed, 96, 32)
    n = rays.shape[0]
    rgb_train = synth.hash_uniform(seed, 800, (n, 3))
    lr_factor = 0.1 ** (1 / 30000)
    opt = torch.optim.Adam(field.get_optparam_groups(0.02, 0.001), betas=(0.9, 0.99))
    names = [k for k in params]
    out = {}
    real = torch.rand_like
    for it in range(steps):
        U = synth.hash_uniform(seed, 810 + it, (n, 1))
        white = it % 2 == 0                                  # iteration 1: black background (the coin came up >= 0.5)
        torch.rand_like = lambda *a, **k: torch.from_numpy(U.copy())
        real_rand = torch.rand
        torch.rand = lambda *a, **k: torch.tensor([0.7])
        try:
            o = field(torch.from_numpy(rays), is_train=True, white_bg=white, N_samples=S, iteration=it)
        finally:
            torch.rand_like = real
            torch.rand = real_rand
        rgb_loss = torch.mean((o["rgb_map"] - torch.from_numpy(rgb_train)) ** 2)
        total = rgb_loss + 8e-5 * field.density_L1()
        opt.zero_grad()
        total.backward()
        for k in names:
            obj = field
            for part in k.split(".")[:-1]:
                obj = getattr(obj, part) if not part.isdigit() else obj[int(part)]
            out[f"grad{it}.{k}"] = getattr(obj, k.split(".")[-1]).grad.numpy().copy()
        out[f"jitter{it}"] = U[:, 0]
        out[f"white{it}"] = np.array(int(white))
        out[f"rgb_map{it}"] = o["rgb_map"].detach().numpy().copy()
        out[f"rgb_loss{it}"] = np.float64(rgb_loss.item())
        out[f"total_loss{it}"] = np.float64(total.item())
        opt.step()
        for g in opt.param_groups:
            g["lr"] = g["lr"] * lr_factor
    for k in names:
        obj = field
        for part in k.split(".")[:-1]:
            obj = getattr(obj, part) if not part.isdigit() else obj[int(part)]
        out[f"after.{k}"] = getattr(obj, k.split(".")[-1]).detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), model="triplane", seed=seed, preset="R1", gauge_std=np.float32(0.05), S=S,
                        steps=steps, aabb=aabb.numpy(), grid=np.array(grid), plane_hw=np.array(plane_hw), gauge_hw=np.array(gauge_hw),
                        near_far=np.array([2.0, 6.0], np.float32), distance_scale=np.float32(25), thr=np.float32(1e-4),
                        stepSize=field.stepSize.numpy(), rays=rays, rgb_train=rgb_train, lr_factor=np.float64(lr_factor),
                        **_checksums(params), **out)
    print(f"{name}: rays {n} S {S} losses {[out[f'rgb_loss{i}'] for i in range(steps)]} |grad plane_xy| {np.abs(out['grad0.plane_xy']).mean():.3e} "
          f"|grad gauge_xy| {np.abs(out['grad0.gauge_xy']).mean():.3e}")


def capture_uv_edit(name="uv_edit", seed=61, n=160):
    """UV-Mapping texture editing: the reference TextureMlpDecoder itself with ``cubemap_`` / ``cubemap_mode_`` set
    (decoder.py:79-121), for a sphere and a square model and all five modes.  ``orig`` (= color1 + color2, which the
    decoder does not return) is taken from the same sub-modules the decoder calls."""
    sys.path.insert(0, os.path.join(REF, "UV-Mapping"))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            dec = importlib.import_module("model.decoder")
            util = importlib.import_module("util")
    finally:
        sys.path.pop(0)
    out = {}
    for prim, uv_dim in (("sphere", 3), ("square", 2)):
        params = synth.uvmapping_params(seed, prim)
        with contextlib.redirect_stdout(io.StringIO()):
            tex = dec.TextureMlpDecoder(3, 10, 6, uv_dim=uv_dim, layers=[5, 3], width=256, clamp=False, primitive_type=prim,
                                        target_texture="None")
        sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
        tex.load_state_dict({k[len("net_texture."):]: v for k, v in sd.items() if k.startswith("net_texture.")})
        q = synth.hash_normal(seed, 900 + uv_dim, (n, 3))
        if prim == "sphere":
            uv = q / np.linalg.norm(q, axis=1, keepdims=True)
            uv[:6] = np.eye(3, dtype=np.float32)[[0, 0, 1, 1, 2, 2]] * np.array([1, -1, 1, -1, 1, -1], np.float32)[:, None]   # face centres
            uv[6] = [0.5, 0.5, 0.70710678]; uv[7] = [0.6, 0.6, -0.52915026]; uv[8] = [-0.57735027] * 3                        # ties / near ties
            cube = synth.hash_uniform(seed, 910, (6, 9, 9, 3))
            cube[:, :3, :3, 0] = 0.995                      # texels above the 0.99 mask threshold of modes 1 / 2
            cube[2, 5:, 5:] = 0.001                         # texels below the 0.01 mask threshold of mode 3
        else:
            uv = np.tanh(q)[:, :2].astype(np.float32)
            uv[:4] = [[-1, -1], [1, 1], [-1, 1], [0.999, -0.3]]
            cube = synth.hash_uniform(seed, 911, (7, 11, 4))
            cube[:2, :4, 0] = 0.995
            cube[4:, 6:] = 0.001
        uv = uv.astype(np.float32)
        view = synth.hash_normal(seed, 920 + uv_dim, (n, 3))
        view = (view / np.linalg.norm(view, axis=1, keepdims=True)).astype(np.float32)
        with torch.no_grad():
            tuv, tv = torch.from_numpy(uv[None, :, None, :uv_dim].copy()), torch.from_numpy(view[None, :, None, :].copy())
            h = tex.block1(torch.cat([tuv, util.positional_encoding(tuv, 10)], -1))
            c1 = torch.nn.functional.softplus(tex.color1(h))
            c2 = tex.block2(torch.cat([h, tv, util.positional_encoding(tv, 6)], -1))
            orig = (c1 + c2)[0, :, 0].numpy()
            # widen the range so that the clamps of every mode are exercised
            orig = (orig * np.float32(3.0) - np.float32(0.4)).astype(np.float32)
            plain = tex(tuv, tv)[0, :, 0].numpy()
            out[f"{prim}.plain"] = plain
            out[f"{prim}.plain_orig"] = (c1 + c2)[0, :, 0].numpy()
            tex.cubemap_ = torch.from_numpy(cube.copy())
            # the decoder recomputes original_color = softplus(color1(h)) + block2(...) from its MLPs; to feed it the widened
            # values the two heads are replaced by constant modules: color1 -> 0, block2 -> orig - softplus(0); what the decoder
            # then sees (softplus(0) + (orig - softplus(0)) in float32) is what the fixture stores as `orig`
            ln2 = torch.nn.functional.softplus(torch.zeros(()))
            c2p = torch.from_numpy(orig.copy()) - ln2
            orig = (ln2 + c2p).numpy().copy()

            class Const(torch.nn.Module):
                def __init__(self, value):
                    super().__init__()
                    self.value = value

                def forward(self, z):
                    return self.value

            real_c1, real_b2 = tex.color1, tex.block2
            tex.color1, tex.block2 = Const(torch.zeros((1, n, 1, 3))), Const(c2p[None, :, None, :])
            try:
                for mode in range(5):
                    tex.cubemap_mode_ = mode
                    out[f"{prim}.mode{mode}"] = tex(tuv, tv)[0, :, 0].numpy().copy()
            finally:
                tex.color1, tex.block2 = real_c1, real_b2
        out[f"{prim}.uv"] = np.concatenate([uv, np.zeros((n, 3 - uv.shape[1]), np.float32)], 1) if uv.shape[1] < 3 else uv
        out[f"{prim}.orig"] = orig
        out[f"{prim}.tex"] = cube
    np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, **out)
    print(f"{name}: " + ", ".join(f"{k} {v.shape}" for k, v in out.items() if k.endswith("mode3")))


def capture_ref_checkpoint(name="ref_ckpt_triplane", seed=71):
    """A checkpoint WRITTEN BY THE REFERENCE (Base.save, FieldBase.py:94-109) for a small model with an alpha mask, plus the
    reference's own render of a few rays with it: the drop-in must load the file as is (main.py:34-38) and reproduce the pixels."""
    F = _import_ref("TriPlane")
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    grid = [12, 10, 9]
    plane_hw = ((10, 12), (9, 10), (9, 12))
    gauge_hw = (6, 7)
    params = synth.triplane_params(seed, plane_hw, gauge_hw, preset="R1", gauge_std=0.04)
    with contextlib.redirect_stdout(io.StringIO()):
        field = F.TriPlane(aabb, grid, "cpu", near_far=[2.0, 6.0], alphaMask_thres=1e-4, distance_scale=25,
                           rayMarch_weight_thres=1e-4, step_ratio=0.5, gauge_start=0)
    _load_params(field, params)
    dhw = (7, 8, 9)
    vol, _ = synth.alpha_mask_bits(seed, dhw)
    field.alphaMask = F.AlphaGridMask("cpu", torch.tensor([[-1.45, -1.4, -1.5], [1.4, 1.5, 1.3]]), torch.from_numpy(vol.astype(np.float32)))
    path = os.path.join(HERE, name + ".th")
    field.save(path)
    rays = _rays_for_case(seed, 64, 32)
    with torch.no_grad():
        out = field(torch.from_numpy(rays), white_bg=True, is_train=False, N_samples=40, iteration=30001)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, rays=rays, rgb_map=out["rgb_map"].numpy(), depth_map=out["depth_map"].numpy(),
                        plane_hw=np.array(plane_hw), gauge_hw=np.array(gauge_hw), mask_dhw=np.array(dhw))
    print(f"{name}: {os.path.getsize(path)} bytes, mean rgb {float(out['rgb_map'].mean()):.4f}")


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1:       # regenerate one fixture without touching the others
        {"evalout": capture_evalout, "train": capture_train, "uv_edit": capture_uv_edit, "ref_ckpt": capture_ref_checkpoint, "rays": capture_rays, "infoinv_alpha": capture_infoinv_alpha}[sys.argv[1]]()
        sys.exit(0)
    capture_ops()
    capture_triplane("triplane_r1_gauge", seed=11, preset="R1", gauge_on=True, gauge_std=0.05, with_mask=False, S=48)
    capture_triplane("triplane_r2_nogauge", seed=12, preset="R2", gauge_on=False, gauge_std=0.05, with_mask=False, S=40,
                     white_bg=False)
    capture_triplane("triplane_r1_mask", seed=13, preset="R1", gauge_on=True, gauge_std=0.02, with_mask=True, S=48)
    capture_triplane("triplane_r0", seed=14, preset="R0", gauge_on=True, gauge_std=0.01, with_mask=False, S=32)
    capture_infoinv("infoinv_r1_on", seed=21, preset="R1", infoinv=True, S=40)
    capture_infoinv("infoinv_r1_off", seed=22, preset="R1", infoinv=False, S=40)
    # round 4 (SURVEY C2's remaining paths): InfoInv WITH an alpha mask on a black background; TriPlane training-mode forwards (supplied
    # per-ray jitter; the background coin below / above 0.5: white / black background with white_bg=False)
    capture_infoinv("infoinv_r1_mask", seed=23, preset="R1", infoinv=True, S=44, with_mask=True, white_bg=False)
    capture_triplane("triplane_r1_train_white", seed=15, preset="R1", gauge_on=True, gauge_std=0.04, with_mask=False, S=44, white_bg=False, train=(0.3,))
    capture_triplane("triplane_r1_train_black", seed=16, preset="R1", gauge_on=True, gauge_std=0.04, with_mask=True, S=44, white_bg=False, train=(0.7,))
    capture_alpha_mask()
    capture_infoinv_alpha()
    capture_uv("uv_sphere", seed=31, primitive_type="sphere")
    capture_uv("uv_square", seed=32, primitive_type="square")
    capture_evalout()
    capture_train()
    capture_uv_edit()
    capture_ref_checkpoint()
    capture_rays()
