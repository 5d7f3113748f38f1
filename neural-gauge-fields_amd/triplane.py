"""``TriPlane`` -- drop-in for the reference's learned-gauge tri-plane field (TriPlane/models/Field.py:13-105)."""
from __future__ import annotations

import torch

from . import _lib
from .fieldbase import AlphaGridMask, Base, renderer, rgb_decoder  # noqa: F401


class TriPlane(Base):
    MODEL = _lib.MODEL_TRIPLANE
    PLANE_C = 64
    DENS_DIM = 16

    def __init__(self, aabb, gridSize, device, **kargs):
        super().__init__(aabb, gridSize, device, **kargs)

    def init_model(self, res=256, dim=64, scale=0.1, device=None, gauge_start=0):
        # parameter set and initialisation of Field.py:17-32
        for name in ('plane_xy', 'plane_yz', 'plane_xz'):
            setattr(self, name, torch.nn.Parameter(scale * torch.randn((1, dim, res, res), device=device)))
        gauge_res = 256
        for name in ('gauge_xy', 'gauge_yz', 'gauge_xz'):
            setattr(self, name, torch.nn.Parameter(torch.zeros((1, 2, gauge_res, gauge_res), device=device)))
        self.rgb_decoder = rgb_decoder(feat_dim=48 * 3, view_pe=2, middle_dim=64).to(device)
        self.density_decoder = torch.nn.Linear(16 * 3, 1).to(device)
        torch.nn.init.xavier_uniform_(self.density_decoder.weight)
        torch.nn.init.constant_(self.density_decoder.bias, 0.0)
        self.gauge_start = gauge_start

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        # Field.py:34-46
        return [{'params': self.plane_xy, 'lr': lr_init_spatialxyz}, {'params': self.plane_yz, 'lr': lr_init_spatialxyz},
                {'params': self.plane_xz, 'lr': lr_init_spatialxyz},
                {'params': self.rgb_decoder.parameters(), 'lr': lr_init_network},
                {'params': self.density_decoder.parameters(), 'lr': lr_init_network},
                {'params': self.gauge_xy, 'lr': lr_init_network * 0.1}, {'params': self.gauge_yz, 'lr': lr_init_network * 0.1},
                {'params': self.gauge_xz, 'lr': lr_init_network * 0.1}]

    def _fill_desc(self, d, dp):
        for k, name in enumerate(('gauge_xy', 'gauge_yz', 'gauge_xz')):
            g = getattr(self, name)
            d.gauge[k] = dp(g)
            d.gauge_h[k], d.gauge_w[k] = g.shape[2], g.shape[3]
        d.dens_w1, d.dens_b1 = dp(self.density_decoder.weight), dp(self.density_decoder.bias)

    def forward(self, rays_chunk, white_bg=True, is_train=False, N_samples=-1, iteration=0, collect_stats=False, out=None):
        """FieldBase.py:251: gauge is applied iff iteration >= gauge_start (Field.py:58)."""
        return self._render(rays_chunk, white_bg, is_train, N_samples, mode=int(iteration >= self.gauge_start),
                            collect_stats=collect_stats, out=out)
