"""Seeded, bit-reproducible synthetic inputs for the ray-march hot path.

Everything here is plain numpy integer/dyadic arithmetic, so the GPU box regenerates
exactly the tensors the golden fixtures were captured with -- no dataset, no checkpoint,
no ``torch.manual_seed`` (whose stream differs between devices and torch versions).

* ``hash_normal`` / ``hash_uniform``: counter-based generator (splitmix64 finaliser over
  ``(seed, stream, index)``); a "normal" is the sum of four 16-bit uniforms, centred and
  scaled to unit variance (Irwin-Hall n=4).  All intermediate values are dyadic rationals
  that float64 holds exactly, so results are identical on every IEEE machine.
* ``triplane_params`` / ``infoinv_params``: the parameter set of the reference modules
  (TriPlane/models/Field.py:17-32, InfoInv/models/Field.py:14-24, networks.py:12-32,34-54)
  as a dict of float32 numpy arrays keyed by the reference's ``state_dict`` names.
* ``lookat_rays``: the Blender-convention pin-hole frame of SURVEY.md section 8 D2
  (TriPlane/dataLoader/ray_utils.py:24-42,66-87 and blender.py:46-53,68,84-85).
* ``edge_rays``: rays that miss the box, start inside it, or have zero direction components.
"""
from __future__ import annotations

import math
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _stream_base(seed: int, stream: int) -> np.uint64:
    s = np.array([(int(seed) * 0x100000001B3 + int(stream) * 0x9E3779B1 + 0x1234567) & 0xFFFFFFFFFFFFFFFF],
                 dtype=np.uint64)
    return _splitmix(_splitmix(s))[0]


def _hash(seed: int, stream: int, n: int) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix(idx * np.uint64(0xD1342543DE82EF95) + _stream_base(seed, stream))


def hash_uniform(seed: int, stream: int, shape) -> np.ndarray:
    """U[0,1) on a 2^-24 lattice, float32 (exactly representable)."""
    n = int(np.prod(shape))
    h = _hash(seed, stream, n)
    u = (h >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)
    return u.astype(np.float32).reshape(shape)


def hash_normal(seed: int, stream: int, shape) -> np.ndarray:
    """Approximately N(0,1): centred sum of four 16-bit uniforms, float32."""
    n = int(np.prod(shape))
    h = _hash(seed, stream, n)
    m = np.uint64(0xFFFF)
    s = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + ((h >> np.uint64(48)) & m))
    z = (s.astype(np.float64) * (1.0 / 65536.0) - 2.0) * math.sqrt(3.0)
    return z.astype(np.float32).reshape(shape)


def _linear(seed, stream, out_f, in_f, bias=True):
    """nn.Linear default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
    bound = np.float32(1.0 / math.sqrt(in_f))
    w = (hash_uniform(seed, stream, (out_f, in_f)) * np.float32(2.0) - np.float32(1.0)) * bound
    b = None
    if bias:
        b = (hash_uniform(seed, stream + 1, (out_f,)) * np.float32(2.0) - np.float32(1.0)) * bound
    return w.astype(np.float32), (None if b is None else b.astype(np.float32))


# Density presets of SURVEY.md section 8 D2: (bias of the last density layer, weight scale).
PRESETS = {"R0": (0.0, 1.0), "R1": (10.0, 20.0), "R2": (6.0, 20.0)}


def triplane_params(seed=0, plane_hw=((256, 256), (256, 256), (256, 256)), gauge_hw=(256, 256),
                    preset="R1", gauge_std=0.01, plane_std=0.1, dim=64):
    """TriPlane parameter dict (state_dict names).  plane_hw = (H,W) of plane_xy, plane_yz, plane_xz."""
    bias, wscale = PRESETS[preset]
    p = {}
    for k, name in enumerate(("xy", "yz", "xz")):
        h, w = plane_hw[k]
        p[f"plane_{name}"] = hash_normal(seed, 10 + k, (1, dim, h, w)) * np.float32(plane_std)
        p[f"gauge_{name}"] = hash_normal(seed, 20 + k, (1, 2, gauge_hw[0], gauge_hw[1])) * np.float32(gauge_std)
    feat = 3 * (dim - 16)
    p["rgb_decoder.basis.weight"], _ = _linear(seed, 30, feat, feat, bias=False)
    p["rgb_decoder.mlp.0.weight"], p["rgb_decoder.mlp.0.bias"] = _linear(seed, 32, 64, feat + 15)
    p["rgb_decoder.mlp.2.weight"], p["rgb_decoder.mlp.2.bias"] = _linear(seed, 34, 64, 64)
    p["rgb_decoder.mlp.4.weight"], _ = _linear(seed, 36, 3, 64, bias=False)
    p["rgb_decoder.mlp.4.bias"] = np.zeros((3,), np.float32)
    # xavier-uniform Linear(48,1), zero bias (Field.py:29-30), then the preset's scale / bias
    bound = np.float32(math.sqrt(6.0 / (48 + 1)))
    wd = (hash_uniform(seed, 40, (1, 48)) * np.float32(2.0) - np.float32(1.0)) * bound
    p["density_decoder.weight"] = (wd * np.float32(wscale)).astype(np.float32)
    p["density_decoder.bias"] = np.full((1,), bias, np.float32)
    return p


def infoinv_params(seed=0, plane_hw=((256, 256), (256, 256), (256, 256)), preset="R1", plane_std=0.1, dim=96):
    """InfoInv parameter dict (state_dict names): 96-ch planes, density MLP 72-32-32-1, rgb_decoder(216)."""
    bias, wscale = PRESETS[preset]
    p = {}
    for k, name in enumerate(("xy", "yz", "xz")):
        h, w = plane_hw[k]
        p[f"plane_{name}"] = hash_normal(seed, 110 + k, (1, dim, h, w)) * np.float32(plane_std)
    dd = 24
    feat = 3 * (dim - dd)
    p["density_decoder.mlp.0.weight"], p["density_decoder.mlp.0.bias"] = _linear(seed, 120, 32, 3 * dd)
    p["density_decoder.mlp.2.weight"], p["density_decoder.mlp.2.bias"] = _linear(seed, 122, 32, 32)
    w3, _ = _linear(seed, 124, 1, 32, bias=False)
    p["density_decoder.mlp.4.weight"] = (w3 * np.float32(wscale)).astype(np.float32)
    p["density_decoder.mlp.4.bias"] = np.full((1,), bias, np.float32)
    p["rgb_decoder.basis.weight"], _ = _linear(seed, 130, feat, feat, bias=False)
    p["rgb_decoder.mlp.0.weight"], p["rgb_decoder.mlp.0.bias"] = _linear(seed, 132, 64, feat + 15)
    p["rgb_decoder.mlp.2.weight"], p["rgb_decoder.mlp.2.bias"] = _linear(seed, 134, 64, 64)
    p["rgb_decoder.mlp.4.weight"], _ = _linear(seed, 136, 3, 64, bias=False)
    p["rgb_decoder.mlp.4.bias"] = np.zeros((3,), np.float32)
    return p


def alpha_mask_bits(seed, shape_dhw, keep=0.6):
    """A seeded {0,1} occupancy volume [D,H,W] (z,y,x) and its np.packbits image (FieldBase.py:104-108)."""
    d, h, w = shape_dhw
    vol = (hash_uniform(seed, 200, (d, h, w)) < np.float32(keep))
    # make it blobby rather than salt-and-pepper: AND with a coarse pattern
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    coarse = ((zz // 3 + yy // 3 + xx // 3) % 4) != 0
    vol = np.logical_and(vol, coarse)
    return vol, np.packbits(vol.reshape(-1))


def lookat_pose(radius=4.0311, elev_deg=30.0, azim_deg=40.0):
    """c2w [3,4] in OpenCV axes (x right, y down, z forward) looking at the origin (blender.py:26,68)."""
    el, az = math.radians(elev_deg), math.radians(azim_deg)
    eye = np.array([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az),
                    radius * math.sin(el)], np.float64)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.stack([right, down, fwd, eye], axis=1)
    return c2w.astype(np.float32)


def lookat_rays(H=800, W=800, c2w=None, camera_angle_x=0.6911112070083618, rows=None):
    """rays [H*W, 6] = (origin, direction), row-major over (row j, col i); float32 arithmetic as the loader.

    rows=(r0, r1) generates only image rows [r0, r1) (what one rank of a sharded render owns).
    """
    if c2w is None:
        c2w = lookat_pose()
    c2w = np.asarray(c2w, np.float32)
    # blender.py:46-47 computes the focal length in double; torch rounds it to float32 once when it divides the pixel grid by it
    focal = np.float32(0.5 * 800 / math.tan(0.5 * camera_angle_x) * (W / 800))
    r0, r1 = (0, H) if rows is None else rows
    i = (np.arange(W, dtype=np.float32) + np.float32(0.5))[None, :].repeat(r1 - r0, 0)
    j = (np.arange(r0, r1, dtype=np.float32) + np.float32(0.5))[:, None].repeat(W, 1)
    dirs = np.stack([(i - np.float32(W / 2)) / focal, (j - np.float32(H / 2)) / focal, np.ones_like(i)], -1)
    dirs = dirs / np.sqrt(np.sum(dirs * dirs, -1, keepdims=True, dtype=np.float32))
    rays_d = (dirs.reshape(-1, 3) @ c2w[:3, :3].T).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([rays_o, rays_d], 1), dtype=np.float32)


def edge_rays(seed=0, n=64, aabb=((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5))):
    """Rays covering the reference's corner cases: miss the box, start inside it, d==0 components."""
    lo, hi = np.asarray(aabb[0], np.float32), np.asarray(aabb[1], np.float32)
    u = hash_uniform(seed, 300, (n, 6))
    o = (u[:, :3] * np.float32(2) - np.float32(1)) * np.float32(4.0)
    tgt = (hash_uniform(seed, 301, (n, 3)) * np.float32(2) - np.float32(1)) * np.float32(1.2)
    d = tgt - o
    k = n // 4
    # quarter 1: start inside the box
    o[:k] = lo + (hi - lo) * u[:k, 3:6]
    d[:k] = (hash_uniform(seed, 302, (k, 3)) * np.float32(2) - np.float32(1))
    # quarter 2: axis-aligned directions (zero components)
    for r in range(k, 2 * k):
        ax = r % 3
        d[r] = 0
        d[r, ax] = -1.0 if o[r, ax] > 0 else 1.0
        if r % 2:
            d[r, (ax + 1) % 3] = 0.3
    # quarter 3: pointing away from the box
    d[2 * k:3 * k] = o[2 * k:3 * k]
    d = d / np.maximum(np.sqrt(np.sum(d * d, -1, keepdims=True)), np.float32(1e-12))
    return np.ascontiguousarray(np.concatenate([o, d], 1), dtype=np.float32)


def _xavier(seed, stream, out_f, in_f, gain=1.0, bias_std=0.0):
    """UV-Mapping's xavier-uniform (util.py:385-395): U(-a, a), a = gain*sqrt(2/(fan_in+fan_out))*sqrt(3);
    biases are zero in the reference (util.py:408-409); bias_std > 0 makes the synthetic test stricter."""
    a = np.float32(gain * math.sqrt(2.0 / (in_f + out_f)) * math.sqrt(3.0))
    w = (hash_uniform(seed, stream, (out_f, in_f)) * np.float32(2.0) - np.float32(1.0)) * a
    b = hash_normal(seed, stream + 1, (out_f,)) * np.float32(bias_std)
    return w.astype(np.float32), b.astype(np.float32)


def uvmapping_params(seed=0, primitive_type="sphere", bias_std=0.05):
    """NeuTex colour-path parameters keyed by the reference's state_dict names:
    net_geometry_decoder.block.{0..22}, gauge_transform.encoder.*, net_texture.{block1,color1,block2}.*"""
    ud = 3 if primitive_type == "sphere" else 2
    g_relu, g_lrelu = math.sqrt(2.0), math.sqrt(2.0 / (1 + 0.2 ** 2))
    p = {}
    st = 500

    def put(prefix, out_f, in_f, gain):
        nonlocal st
        p[prefix + ".weight"], p[prefix + ".bias"] = _xavier(seed, st, out_f, in_f, gain, bias_std)
        st += 2

    put("net_geometry_decoder.block.0", 256, 63, g_relu)
    for i in range(10):
        put(f"net_geometry_decoder.block.{2 + 2 * i}", 256, 256, g_relu)
    put("net_geometry_decoder.block.22", 1, 256, 1.0)
    put("gauge_transform.encoder.linear1", 64, 63, 1.0)
    put("gauge_transform.encoder.linear2", 128, 64, 1.0)
    put("gauge_transform.encoder.linear_list.0", 128, 128, 1.0)
    put("gauge_transform.encoder.linear_list.1", 128, 128, 1.0)
    put("gauge_transform.encoder.last_linear", ud, 128, 1.0)
    put("net_texture.block1.0", 256, ud + 20 * ud, g_lrelu)
    for i in range(5):
        put(f"net_texture.block1.{2 + 2 * i}", 256, 256, g_lrelu)
    put("net_texture.color1", 3, 256, 1.0)
    put("net_texture.block2.0", 256, 295, g_lrelu)
    for i in range(3):
        put(f"net_texture.block2.{2 + 2 * i}", 256, 256, g_lrelu)
    put("net_texture.block2.8", 3, 256, 1.0)
    return p


def dtu_rays(H=600, W=800, campos=(1.7514, 0.0961, -4.7220), focal=(1446.17, 1441.59), center=(411.60, 309.54), rows=None):
    """Pin-hole rays of a DTU-like view (UV-Mapping/data/dtu.py:27-37 convention: normalised directions through
    pixel centres, camera looking at the origin).  Returns campos [3] and raydir [rows*W, 3] float32."""
    c = np.asarray(campos, np.float64)
    fwd = -c / np.linalg.norm(c)
    up = np.array([0.0, -1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    r0, r1 = (0, H) if rows is None else rows
    i = (np.arange(W, dtype=np.float64) + 0.5)[None, :].repeat(r1 - r0, 0)
    j = (np.arange(r0, r1, dtype=np.float64) + 0.5)[:, None].repeat(W, 1)
    d = ((i - center[0]) / focal[0])[..., None] * right + ((j - center[1]) / focal[1])[..., None] * down + fwd
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return c.astype(np.float32), np.ascontiguousarray(d.reshape(-1, 3).astype(np.float32))


# Camera 0 of the DTU scan the reference ships (UV-Mapping/data/DTU/scan83/trainData/in_camFocal.npy, in_camPrincpt.npy,
# in_camExtrinsics.npy[0][:3,:3], in_camOrgs.npy[0]; float32 values, also stored in tests/golden/rays_dtu.npz): the view of
# BASELINE config 4's bench leg.  The extrinsic block is a scaled rotation (|row| ~ 135.9), as shipped.
DTU_VIEW0 = {
    "focal": (1446.165283203125, 1441.587646484375),
    "princpt": (411.6026611328125, 309.53546142578125),
    "rot": ((131.82720947265625, 1.0162771940231323, 32.87162780761719),
            (-2.003077507019043, 135.79864501953125, 3.834646224975586),
            (-32.82627487182617, -4.205236434936523, 131.7753448486328)),
    "campos": (1.7514456510543823, 0.09612564742565155, -4.722002983093262),
}


def dtu_rays_dir(H, W, focal, princpt, rot, rows=None):
    """Restatement of get_rays_dir on the 'no_crop' pixel grid (UV-Mapping/data/dtu.py:27-37,160-168), float32 like the
    reference (float32 pixel coordinates, focal, principal point and rotation): raydir [rows*W, 3]."""
    f, c, r = np.asarray(focal, np.float32), np.asarray(princpt, np.float32), np.asarray(rot, np.float32)
    r0, r1 = (0, H) if rows is None else rows
    px = np.arange(W, dtype=np.float32)[None, :].repeat(r1 - r0, 0)
    py = np.arange(r0, r1, dtype=np.float32)[:, None].repeat(W, 1)
    x, y = (px - c[0]) / f[0], (py - c[1]) / f[1]
    d = np.stack([(r[0, j] * x + r[1, j] * y) + r[2, j] for j in range(3)], -1)         # sum_i rot[i][j] * dirs[i], i = 0, 1, 2 in order
    n = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])
    d = d / (n + np.float32(1e-5))[..., None]
    return np.ascontiguousarray(d.reshape(-1, 3), dtype=np.float32)
