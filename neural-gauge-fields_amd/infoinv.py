"""``TriPlane`` of the InfoInv tree (InfoInv/models/Field.py:10-89): no gauge, 96-channel planes,
plane features modulated by a sinusoidal encoding of the position, density MLP 72-32-32-1."""
from __future__ import annotations

import torch

from . import _lib
from .fieldbase import AlphaGridMask, Base, density_decoder, renderer, rgb_decoder  # noqa: F401


class TriPlane(Base):
    MODEL = _lib.MODEL_INFOINV
    PLANE_C = 96
    DENS_DIM = 24

    def __init__(self, aabb, gridSize, device, **kargs):
        kargs.pop('gauge_start', None)
        super().__init__(aabb, gridSize, device, **kargs)

    def init_model(self, res=256, dim=96, scale=0.1, device=None, gauge_start=0):
        for name in ('plane_xy', 'plane_yz', 'plane_xz'):
            setattr(self, name, torch.nn.Parameter(scale * torch.randn((1, dim, res, res), device=device)))
        self.density_dim = 24
        self.rgb_dim = dim - self.density_dim
        self.density_decoder = density_decoder(feat_dim=self.density_dim * 3, middle_dim=32).to(device)
        self.rgb_decoder = rgb_decoder(feat_dim=self.rgb_dim * 3, view_pe=2, middle_dim=64).to(device)

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        return [{'params': self.plane_xy, 'lr': lr_init_spatialxyz}, {'params': self.plane_yz, 'lr': lr_init_spatialxyz},
                {'params': self.plane_xz, 'lr': lr_init_spatialxyz},
                {'params': self.rgb_decoder.parameters(), 'lr': lr_init_network},
                {'params': self.density_decoder.parameters(), 'lr': lr_init_network}]

    def _fill_desc(self, d, dp):
        m = self.density_decoder.mlp
        d.dens_w1, d.dens_b1 = dp(m[0].weight), dp(m[0].bias)
        d.dens_w2, d.dens_b2 = dp(m[2].weight), dp(m[2].bias)
        d.dens_w3, d.dens_b3 = dp(m[4].weight), dp(m[4].bias)

    def _alpha_mode(self, infoinv=True) -> int:
        """compute_alpha / getDenseAlpha / updateAlphaMask(..., infoinv=True) of InfoInv/models/FieldBase.py:140,161,180."""
        return int(bool(infoinv))

    def forward(self, rays_chunk, white_bg=True, is_train=False, N_samples=-1, infoinv=True, collect_stats=False, out=None, jitter=None, coin=None, row_width=0):
        """InfoInv/models/FieldBase.py:228.  Training the InfoInv tree is outside SURVEY.md section 8: a call that the
        reference would record for autograd (is_train=True, grad enabled, parameters requiring gradients) raises instead of returning pixels
        without a graph -- a loss built on them would otherwise only train its regularisers."""
        if self._wants_grad(is_train):
            raise NotImplementedError("ngf_amd.infoinv.TriPlane has no backward (InfoInv training is out of scope, DESIGN.md section 7): call it "
                                      "under torch.no_grad() for a training-mode render, or train with the TriPlane tree")
        return self._render(rays_chunk, white_bg, is_train, N_samples, mode=int(bool(infoinv)), collect_stats=collect_stats, out=out, jitter=jitter, coin=coin, row_width=row_width)
