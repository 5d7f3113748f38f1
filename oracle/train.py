"""CPU oracle of ONE training step of the TriPlane model -- TEST INFRASTRUCTURE ONLY
(tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product path never imports this).

What it restates (reference file:line):
    forward in training mode   Base.forward(is_train=True)        TriPlane/models/FieldBase.py:251-312
                               sample_ray jitter                  FieldBase.py:127-131 (one U[0,1) per ray, an INPUT here)
                               random white background            FieldBase.py:299 (the coin is an INPUT here: ``white_bg``)
    loss                       mean((rgb_map-rgb_train)**2) + 8e-5 * density_L1()     main.py:281-291, Field.py:149-152
    backward                   torch autograd through the eager port (oracle/eager.py) -- the reference's arithmetic lives
                               in ATen, so autograd of the same operator sequence IS the reference gradient
    optimiser                  torch.optim.Adam(betas=(0.9, 0.99)) over the 8 groups of get_optparam_groups
                               (Field.py:34-46), lr *= lr_factor after every step (main.py:298-299); restated in
                               ``adam_update`` (plain tensor arithmetic) and pinned against torch.optim.Adam in the tests.
Pinned by tests/golden/train_r1.npz (gradients and two Adam steps captured from the reference module itself).
"""
from __future__ import annotations

import math

import torch

from .eager import EagerField

L1_REG_WEIGHT = 8e-5          # main.py:262
PLANES = ("plane_xy", "plane_yz", "plane_xz")
GAUGES = ("gauge_xy", "gauge_yz", "gauge_xz")


def param_groups(params, lr_init_spatialxyz=0.02, lr_init_network=0.001):
    """get_optparam_groups (Field.py:34-46): [(names, lr)] in the reference's order."""
    rgb = [k for k in params if k.startswith("rgb_decoder.")]
    dens = [k for k in params if k.startswith("density_decoder.")]
    return [(["plane_xy"], lr_init_spatialxyz), (["plane_yz"], lr_init_spatialxyz), (["plane_xz"], lr_init_spatialxyz),
            (rgb, lr_init_network), (dens, lr_init_network),
            (["gauge_xy"], lr_init_network * 0.1), (["gauge_yz"], lr_init_network * 0.1), (["gauge_xz"], lr_init_network * 0.1)]


class EagerTrainer(EagerField):
    """EagerField with differentiable parameters and the training-mode forward."""

    def __init__(self, params, aabb, step, near_far=(2.0, 6.0), distance_scale=25.0, thr=1e-4, gauge_start=0, alpha_mask=None):
        super().__init__(params, aabb, step, near_far, distance_scale, thr, "triplane", alpha_mask)
        self.p = {k: v.clone().requires_grad_(True) for k, v in self.p.items()}
        self.gauge_start = gauge_start

    def forward_train(self, rays, S, jitter, white_bg, iteration):
        """Base.forward(is_train=True): jitter [n] in [0,1) (FieldBase.py:129-130), white_bg already includes the coin."""
        o, d = rays[:, :3], rays[:, 3:6]
        vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
        tmin = torch.minimum((self.aabb[1] - o) / vec, (self.aabb[0] - o) / vec).amax(-1).clamp(min=self.near, max=self.far)
        rng = torch.arange(S)[None].float().repeat(rays.shape[0], 1) + jitter.reshape(-1, 1)
        z = tmin[:, None] + self.step * rng
        pts = o[:, None, :] + d[:, None, :] * z[..., None]
        valid = ~((self.aabb[0] > pts) | (pts > self.aabb[1])).any(-1)
        dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), -1)
        if self.mask is not None:                                   # FieldBase.py:261-267
            vol, maabb, minv = self.mask
            q = (pts[valid] - maabb[0]) * minv - 1
            a = torch.nn.functional.grid_sample(vol, q.reshape(1, -1, 1, 1, 3), align_corners=True).reshape(-1)
            bad = ~valid
            bad[valid] |= ~(a > 0)
            valid = ~bad
        n = rays.shape[0]
        sigma = torch.zeros((n, S))
        coords = [torch.zeros((n, S, 2)) for _ in range(3)]
        gauge_on = iteration >= self.gauge_start
        if valid.any():
            x = (pts - self.aabb[0]) * self.inv - 1
            c = self._coords(x[valid], gauge_on)
            sigma = sigma.masked_scatter(valid, self._sigma(c, True))
            coords = [ck.masked_scatter(valid[..., None].expand(n, S, 2), cv) for ck, cv in zip(coords, c)]
        alpha = 1.0 - torch.exp(-sigma * (dists * self.dscale))
        T = torch.cumprod(torch.cat([torch.ones(n, 1), 1.0 - alpha + 1e-10], -1), -1)
        w = alpha * T[:, :-1]
        act = w > self.thr
        rgb = torch.zeros((n, S, 3))
        if act.any():
            dirs = d[:, None, :].expand(n, S, 3)
            rgb = rgb.masked_scatter(act[..., None].expand(n, S, 3), self._rgb([ck[act] for ck in coords], dirs[act], True))
        acc = w.sum(-1)
        rgb_map = (w[..., None] * rgb).sum(-2)
        if white_bg:
            rgb_map = rgb_map + (1.0 - acc[..., None])
        return rgb_map.clamp(0, 1), {"weight": w, "sigma": sigma, "active": act}

    def loss(self, rays, rgb_train, S, jitter, white_bg, iteration):
        rgb_map, aux = self.forward_train(rays, S, jitter, white_bg, iteration)
        rgb_loss = torch.mean((rgb_map - rgb_train) ** 2)
        l1 = sum(torch.mean(torch.abs(self.p[k])) for k in PLANES)          # density_L1, Field.py:149-152
        return rgb_loss + L1_REG_WEIGHT * l1, rgb_loss, rgb_map, aux

    def gradients(self, rays, rgb_train, S, jitter, white_bg, iteration):
        for v in self.p.values():
            v.grad = None
        total, rgb_loss, rgb_map, aux = self.loss(rays, rgb_train, S, jitter, white_bg, iteration)
        total.backward()
        return {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in self.p.items()}, float(rgb_loss.detach()), rgb_map.detach(), aux


def adam_update(p, g, m, v, t, lr, beta1=0.9, beta2=0.99, eps=1e-8):
    """torch.optim.Adam's single-tensor update (no weight decay, no amsgrad), float32: returns (p, m, v) after step t>=1."""
    m = m + (1 - beta1) * (g - m)              # exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + g * g * (1 - beta2)
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    step_size = lr / bc1
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - step_size * (m / denom), m, v
