"""PyTorch-eager CPU restatement of the reference path -- TEST / BASELINE INFRASTRUCTURE ONLY.

This is what bench.py times as ``cpu_baseline`` (kind "port"): the same ATen operators, in the same
order, as the reference's ``Base.forward`` (TriPlane/models/FieldBase.py:251-312, Field.py:53-105,
networks.py:25-32; InfoInv/models/Field.py:52-89), driven chunk by chunk like ``renderer``
(TriPlane/main.py:60-71).  The reference itself cannot travel to the GPU box, so its eager cost is
measured through this port; tests/test_eager_port.py checks it against the golden vectors captured
from the reference and against the C oracle.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _sample2d(plane, uv):
    """[1,C,H,W] sampled at uv [M,2] -> [M,C] (bilinear, zeros padding, align_corners=True)."""
    return F.grid_sample(plane, uv.reshape(1, -1, 1, 2), align_corners=True).reshape(plane.shape[1], -1).T


def _posenc(x, freqs):
    bands = 2.0 ** torch.arange(freqs, dtype=torch.float32)
    y = (x[..., None] * bands).reshape(*x.shape[:-1], -1)
    return torch.cat([torch.sin(y), torch.cos(y)], -1)


class EagerField:
    def __init__(self, params: dict, aabb, step, near_far=(2.0, 6.0), distance_scale=25.0, thr=1e-4,
                 model="triplane", alpha_mask=None):
        self.p = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in params.items()}
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).reshape(2, 3)
        self.inv = 2.0 / (self.aabb[1] - self.aabb[0])
        self.step = torch.tensor(float(step), dtype=torch.float32)
        self.near, self.far = float(near_far[0]), float(near_far[1])
        self.dscale, self.thr = float(distance_scale), float(thr)
        self.infoinv_model = model == "infoinv"
        self.dd = 24 if self.infoinv_model else 16
        self.mask = None
        if alpha_mask is not None:
            vol, maabb = alpha_mask                      # float volume [D,H,W], aabb [2,3]
            maabb = torch.as_tensor(maabb, dtype=torch.float32)
            self.mask = (torch.as_tensor(vol, dtype=torch.float32)[None, None], maabb, 1.0 / (maabb[1] - maabb[0]) * 2)

    # -- pieces -------------------------------------------------------------------------------------
    def _coords(self, x, gauge_on):
        xy, yz, xz = x[:, :2], x[:, 1:], x[:, ::2]
        if self.infoinv_model or not gauge_on:
            return xy, yz, xz
        dxy = _sample2d(self.p["gauge_xy"], xy)
        dyz = _sample2d(self.p["gauge_yz"], yz)
        dxz = _sample2d(self.p["gauge_xz"], xz)
        txy, tyz, txz = xy + dxy, yz + dyz, xz + dxz
        txy = torch.stack([txy[:, 0] + dxz[:, 0], txy[:, 1] + dyz[:, 0]], -1)
        tyz = torch.stack([tyz[:, 0] + dxy[:, 1], tyz[:, 1] + dxz[:, 1]], -1)
        txz = torch.stack([txz[:, 0] + dxy[:, 0], txz[:, 1] + dyz[:, 1]], -1)
        return txy, tyz, txz

    def _plane_feats(self, c, lo, hi, pe):
        out = []
        for name, uv in zip(("plane_xy", "plane_yz", "plane_xz"), c):
            f = _sample2d(self.p[name][:, lo:hi], uv)
            out.append(f * pe if pe is not None else f)
        return torch.cat(out, -1)

    def _sigma(self, c, modulate):
        if not self.infoinv_model:
            f = self._plane_feats(c, 0, 16, None)
            return F.softplus(F.linear(f, self.p["density_decoder.weight"], self.p["density_decoder.bias"]).reshape(-1) - 10)
        pe = _posenc(torch.cat([c[0], c[1][:, 1:]], -1), 4) if modulate else None
        h = self._plane_feats(c, 0, 24, pe)
        for i in (0, 2, 4):
            h = F.linear(h, self.p[f"density_decoder.mlp.{i}.weight"], self.p[f"density_decoder.mlp.{i}.bias"])
            if i < 4:
                h = torch.relu(h)
        return F.softplus(h - 10).reshape(-1)

    def _rgb(self, c, dirs, modulate):
        pe = _posenc(torch.cat([c[0], c[1][:, 1:]], -1), 12) if (self.infoinv_model and modulate) else None
        C = self.p["plane_xy"].shape[1]
        f = self._plane_feats(c, self.dd, C, pe)
        h = torch.cat([F.linear(f, self.p["rgb_decoder.basis.weight"]), dirs, _posenc(dirs, 2)], -1)
        for i in (0, 2, 4):
            h = F.linear(h, self.p[f"rgb_decoder.mlp.{i}.weight"], self.p[f"rgb_decoder.mlp.{i}.bias"])
            if i < 4:
                h = torch.relu(h)
        return torch.sigmoid(h)

    # -- Base.forward ---------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, rays, S, white_bg=True, mode=True):
        o, d = rays[:, :3], rays[:, 3:6]
        vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
        tmin = torch.minimum((self.aabb[1] - o) / vec, (self.aabb[0] - o) / vec).amax(-1).clamp(min=self.near, max=self.far)
        z = tmin[:, None] + self.step * torch.arange(S)[None].float()
        pts = o[:, None, :] + d[:, None, :] * z[..., None]
        valid = ~((self.aabb[0] > pts) | (pts > self.aabb[1])).any(-1)
        dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), -1)
        if self.mask is not None:
            vol, maabb, minv = self.mask
            q = (pts[valid] - maabb[0]) * minv - 1
            a = F.grid_sample(vol, q.reshape(1, -1, 1, 1, 3), align_corners=True).reshape(-1)
            bad = ~valid
            bad[valid] |= ~(a > 0)
            valid = ~bad
        n = rays.shape[0]
        sigma = torch.zeros((n, S))
        coords = [torch.zeros((n, S, 2)) for _ in range(3)]
        if valid.any():
            x = (pts - self.aabb[0]) * self.inv - 1
            c = self._coords(x[valid], mode)
            sigma[valid] = self._sigma(c, mode)
            for k in range(3):
                coords[k][valid] = c[k]
        alpha = 1.0 - torch.exp(-sigma * (dists * self.dscale))
        T = torch.cumprod(torch.cat([torch.ones(n, 1), 1.0 - alpha + 1e-10], -1), -1)
        w = alpha * T[:, :-1]
        act = w > self.thr
        rgb = torch.zeros((n, S, 3))
        if act.any():
            dirs = d[:, None, :].expand(n, S, 3)
            rgb[act] = self._rgb([ck[act] for ck in coords], dirs[act], mode)
        acc = w.sum(-1)
        rgb_map = (w[..., None] * rgb).sum(-2)
        if white_bg:
            rgb_map = rgb_map + (1.0 - acc[..., None])
        rgb_map = rgb_map.clamp(0, 1)
        depth = (w * z).sum(-1) + (1.0 - acc) * rays[..., -1]
        return rgb_map, depth, float(act.float().mean())

    def render(self, rays, S, chunk=4096, white_bg=True, mode=True):
        """The renderer chunk loop (TriPlane/main.py:60-71)."""
        rgbs, depths = [], []
        for i in range(0, rays.shape[0], chunk):
            r, dp, _ = self.forward(rays[i:i + chunk], S, white_bg, mode)
            rgbs.append(r)
            depths.append(dp)
        return torch.cat(rgbs), torch.cat(depths)
