"""CPU oracle of the UV-Mapping texture-editing stage (TEST INFRASTRUCTURE ONLY).

numpy restatement of TextureMlpDecoder.forward's cubemap branch (UV-Mapping/model/decoder.py:95-121) on top of
sample_cubemap / sample_square (UV-Mapping/util.py:172-238, 277-282; F.grid_sample bilinear, padding_mode='border',
align_corners=False).  Pinned by tests/golden/uv_edit.npz: outputs of the reference decoder itself with ``cubemap_`` set.
"""
import numpy as np

f32 = np.float32


def grid_sample_border(tex, u, v):
    """tex [H,W,C]; u -> W, v -> H; one output row per point."""
    H, W, C = tex.shape
    ix = ((u + f32(1)) * f32(W) - f32(1)) / f32(2)
    iy = ((v + f32(1)) * f32(H) - f32(1)) / f32(2)
    ix = np.minimum(f32(W - 1), np.maximum(ix, f32(0)))
    iy = np.minimum(f32(H - 1), np.maximum(iy, f32(0)))
    x0 = np.floor(ix).astype(np.int64)
    y0 = np.floor(iy).astype(np.int64)
    tx = (ix - x0.astype(f32)).astype(f32)
    ty = (iy - y0.astype(f32)).astype(f32)
    out = np.zeros((u.shape[0], C), f32)
    for dx, dy, w in ((0, 0, (1 - tx) * (1 - ty)), (1, 0, tx * (1 - ty)), (0, 1, (1 - tx) * ty), (1, 1, tx * ty)):
        xs, ys = x0 + dx, y0 + dy
        ok = (xs <= W - 1) & (ys <= H - 1)
        out[ok] += tex[ys[ok], xs[ok]] * w[ok, None].astype(f32)
    return out


def sample_cubemap(cube, xyz):
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    px, py, pz = x > 0, y > 0, z > 0
    masks = [px & (ax >= ay) & (ax >= az), ~px & (ax >= ay) & (ax >= az), py & (ay >= ax) & (ay >= az), ~py & (ay >= ax) & (ay >= az),
             pz & (az >= ax) & (az >= ay), ~pz & (az >= ax) & (az >= ay)]
    with np.errstate(divide="ignore", invalid="ignore"):
        uvs = [(-z / ax, y / ax), (z / ax, y / ax), (x / ay, -z / ay), (x / ay, z / ay), (x / az, y / az), (-x / az, y / az)]
    out = np.zeros((xyz.shape[0], cube.shape[-1]), f32)
    for face in range(6):                                   # later faces overwrite earlier ones on ties (util.py:225-236)
        m = masks[face]
        if m.any():
            out[m] = grid_sample_border(cube[face], uvs[face][0][m].astype(f32), uvs[face][1][m].astype(f32))
    return out


def texture_edit(tex, mode, sphere, uv, orig):
    """decoder.py:95-121.  tex [6,R,R,C] | [H,W,C]; uv [n,3]; orig = color1 + color2 [n,3] -> [n,3]."""
    tex, uv, orig = np.asarray(tex, f32), np.asarray(uv, f32), np.asarray(orig, f32)
    cc = sample_cubemap(tex, uv) if sphere else grid_sample_border(tex, uv[:, 0], uv[:, 1])
    if mode == 0:
        o = np.clip(orig * f32(8), 0, 1)
        return (cc * o.mean(-1, keepdims=True))[:, :3].astype(f32)
    o = np.clip(orig, 0, 1).astype(f32)
    if mode == 1:
        m = cc[:, 0] < 0.99
        o[m] *= cc[m][:, :3]
        return o
    if mode == 2:
        m = cc[:, 0] < 0.99
        with np.errstate(divide="ignore", invalid="ignore"):
            o[m] *= (f32(1) / cc[m][:, :3])
        return o
    if mode == 3:
        m = cc[:, :3].sum(-1) > 0.01
        o[m] = f32(2) * o[m].mean(-1)[:, None] * cc[:, :3][m]
        return (o + cc[:, :3]).astype(f32)
    return np.clip(cc[:, :3], 0, 1).astype(f32)
