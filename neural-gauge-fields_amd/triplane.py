"""``TriPlane`` -- drop-in for the reference's learned-gauge tri-plane field (TriPlane/models/Field.py:13-105)."""
from __future__ import annotations

import torch

from . import _lib
from .fieldbase import AlphaGridMask, Base, renderer, rgb_decoder  # noqa: F401


class _DensityL1(torch.autograd.Function):
    """mean|plane_xy| + mean|plane_yz| + mean|plane_xz| (Field.py:149-152) and sign(p) * upstream / numel as its gradient."""

    @staticmethod
    def forward(ctx, pxy, pyz, pxz):
        import ctypes as C
        planes = (pxy.detach(), pyz.detach(), pxz.detach())
        out = torch.empty((), device=pxy.device, dtype=torch.float32)
        ws = torch.empty((3 * 256,), device=pxy.device, dtype=torch.float64)
        ptrs = (C.c_void_p * 3)(*[p.data_ptr() for p in planes])
        ns = (C.c_int64 * 3)(*[p.numel() for p in planes])
        with torch.cuda.device(pxy.device):
            _lib.check(_lib.lib().ngf_planes_l1(ptrs, ns, out.data_ptr(), ws.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ctx.save_for_backward(pxy, pyz, pxz)
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        planes = ctx.saved_tensors
        g = g.to(dtype=torch.float32).contiguous()
        grads = [torch.empty_like(p) if need else None for p, need in zip(planes, ctx.needs_input_grad)]
        ptrs = (C.c_void_p * 3)(*[p.data_ptr() for p in planes])
        ns = (C.c_int64 * 3)(*[p.numel() for p in planes])
        gp = (C.c_void_p * 3)(*[None if x is None else x.data_ptr() for x in grads])
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().ngf_planes_l1_backward(ptrs, ns, g.data_ptr(), gp, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return tuple(grads)


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

    @torch.no_grad()
    def up_sampling(self, res):
        """Field.py:108-114: the three planes are resampled to the new grid (bilinear, align_corners=True) on the device
        through ngf_resize_bilinear; gauge planes keep their size.  A Trainer built on the old tensors must be rebuilt."""
        import ctypes as C
        L = _lib.lib()
        for name, (h, w) in (('plane_xy', (res[1], res[0])), ('plane_yz', (res[2], res[1])), ('plane_xz', (res[2], res[0]))):
            old = getattr(self, name).data.contiguous()
            if not old.is_cuda:
                raise RuntimeError("up_sampling runs on the GPU only; there is no CPU path")
            new = torch.empty((1, old.shape[1], int(h), int(w)), device=old.device, dtype=torch.float32)
            with torch.cuda.device(old.device):
                _lib.check(L.ngf_resize_bilinear(old.data_ptr(), old.shape[1], old.shape[2], old.shape[3], new.data_ptr(), int(h), int(w),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            setattr(self, name, torch.nn.Parameter(new))
        self.init_para(res)

    @torch.no_grad()
    def shrink(self, new_aabb):
        """Field.py:117-132: crop the planes to the grid cells covering new_aabb (pure slicing), then init_para."""
        xyz_min, xyz_max = new_aabb
        aabb0 = self.aabb[0].to(self.units.device)
        t_l = (xyz_min.to(self.units.device) - aabb0) / self.units
        b_r = (xyz_max.to(self.units.device) - aabb0) / self.units
        t_l, b_r = torch.round(torch.round(t_l)).long(), torch.round(b_r).long() + 1
        b_r = torch.stack([b_r, self.gridSize]).amin(0)
        self.plane_xy = torch.nn.Parameter(self.plane_xy.data[..., t_l[1]:b_r[1], t_l[0]:b_r[0]].contiguous())
        self.plane_yz = torch.nn.Parameter(self.plane_yz.data[..., t_l[2]:b_r[2], t_l[1]:b_r[1]].contiguous())
        self.plane_xz = torch.nn.Parameter(self.plane_xz.data[..., t_l[2]:b_r[2], t_l[0]:b_r[0]].contiguous())
        newSize = b_r - t_l
        self.aabb = torch.stack([xyz_min, xyz_max]) if not torch.is_tensor(new_aabb) else new_aabb
        self.init_para((int(newSize[0]), int(newSize[1]), int(newSize[2])))

    def density_L1(self):
        """Field.py:149-152, differentiable like the reference's (the reference loop adds it to the loss, main.py:279-281); ngf_amd.train.Trainer fuses
        its gradient into the planes' Adam kernel instead.  On the device: ngf_planes_l1 / ngf_planes_l1_backward (three planes per launch, no |plane|
        intermediates) behind one autograd node; anything else (CPU tensors, other dtypes) is the reference's torch expression."""
        planes = (self.plane_xy, self.plane_yz, self.plane_xz)
        if all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() % 16 == 0 and p.numel() > 0 for p in planes):
            return _DensityL1.apply(*planes)
        return torch.mean(torch.abs(self.plane_xy)) + torch.mean(torch.abs(self.plane_yz)) + torch.mean(torch.abs(self.plane_xz))

    def _fill_desc(self, d, dp):
        for k, name in enumerate(('gauge_xy', 'gauge_yz', 'gauge_xz')):
            g = getattr(self, name)
            d.gauge[k] = dp(g)
            d.gauge_h[k], d.gauge_w[k] = g.shape[2], g.shape[3]
        d.dens_w1, d.dens_b1 = dp(self.density_decoder.weight), dp(self.density_decoder.bias)

    def forward(self, rays_chunk, white_bg=True, is_train=False, N_samples=-1, iteration=0, collect_stats=False, out=None, jitter=None, coin=None, row_width=0):
        """FieldBase.py:251: gauge is applied iff iteration >= gauge_start (Field.py:58).  With ``is_train=True``, autograd enabled and
        parameters that require gradients -- the reference's training loop, TriPlane/main.py:272 -- the result is differentiable with respect to
        the fifteen parameters (``Base._render_train``); everything else is the fused eval launch (no graph, like the reference under its
        ``@torch.no_grad()`` drivers)."""
        if self._wants_grad(is_train):
            if collect_stats or out is not None:
                raise ValueError("collect_stats / out= belong to the eval launch; wrap the call in torch.no_grad() to use them with is_train=True")
            return self._render_train(rays_chunk, white_bg, N_samples, int(iteration >= self.gauge_start), jitter=jitter, coin=coin)
        return self._render(rays_chunk, white_bg, is_train, N_samples, mode=int(iteration >= self.gauge_start),
                            collect_stats=collect_stats, out=out, jitter=jitter, coin=coin, row_width=row_width)
