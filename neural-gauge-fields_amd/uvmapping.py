"""``NeuTex`` -- drop-in for the colour path of the reference's UV-Mapping model (UV-Mapping/model/model.py:11-59).

Same sub-module and parameter names as the reference (``net_geometry_decoder.block.*``,
``gauge_transform.encoder.*``, ``net_texture.{block1,color1,block2}.*``) so its ``{epoch}_net_NeuTex.pth``
checkpoints load (``strict=False`` skips ``inverse_gauge.*``, which only feeds training losses, model.py:335-350).
``forward`` returns the reference's ``color`` / ``transmittance`` outputs; the three MLPs, the cube ray generation
and the ray march run in one HIP kernel (include/ngf.h: ngf_uv_render).

The reference jitters the segment lengths with ``torch.rand`` even at test time (model.py:30); pass ``jitter_u``
([N,R,S] uniforms) for reproducible output, otherwise they are drawn on the device.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib


def _seq(dims, act):
    layers = []
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        layers.append(nn.Linear(a, b))
        if i < len(dims) - 2 or act[1]:
            layers.append(act[0]())
    return nn.Sequential(*layers)


class GeometryMlpDecoder(nn.Module):
    """decoder.py:201-217 (parameter container; evaluated inside the kernel)."""

    def __init__(self, pos_freqs=10, hidden_size=256, num_layers=10):
        super().__init__()
        if (pos_freqs, hidden_size, num_layers) != (10, 256, 10):
            raise ValueError("the gfx950 kernel implements GeometryMlpDecoder(10, 256, 10) (model.py:16)")
        dims = [3 + 6 * pos_freqs] + [hidden_size] * (num_layers + 1) + [1]
        self.block = _seq(dims, (nn.ReLU, False))


class GaugeNetwork(nn.Module):
    """gauge_fields.py:8-34."""

    def __init__(self, input_dim, output_dim, mid_size=64, hidden_size=128, num_layers=2):
        super().__init__()
        self.linear1 = nn.Linear(input_dim + 2 * input_dim * 10, mid_size)
        self.linear2 = nn.Linear(mid_size, hidden_size)
        self.linear_list = nn.ModuleList([nn.Linear(hidden_size, hidden_size) for _ in range(num_layers)])
        self.last_linear = nn.Linear(hidden_size, output_dim)


class GaugeTransform(nn.Module):
    """gauge_fields.py:49-58."""

    def __init__(self, primitive_type):
        super().__init__()
        self.output_dim = 2 if primitive_type == 'square' else 3
        self.encoder = GaugeNetwork(3, self.output_dim)


class TextureMlpDecoder(nn.Module):
    """decoder.py:11-58: the MLPs; ``cubemap_`` / ``cubemap_mode_`` (texture editing) are plain attributes as in the
    reference and are pushed to the device by NeuTex.set_target_texture."""

    def __init__(self, uv_dim, width=256):
        super().__init__()
        self.cubemap_ = None
        self.cubemap_mode_ = 0
        self.block1 = _seq([uv_dim + 20 * uv_dim] + [width] * 6, (lambda: nn.LeakyReLU(0.2), True))
        self.color1 = nn.Linear(width, 3)
        self.block2 = _seq([width + 3 + 36] + [width] * 4 + [3], (lambda: nn.LeakyReLU(0.2), False))


class NeuTex(nn.Module):
    def __init__(self, opt=None, primitive_type=None, sample_num=None, device='cuda', split_bf16=False):
        super().__init__()
        self.opt = opt
        self.split_bf16 = bool(split_bf16)          # NGF_UV_F_SPLIT_BF16: 256-unit layers as 3-term split bf16 MFMA products (opt-in)
        self.primitive_type = primitive_type or getattr(opt, 'primitive_type', 'square')
        self.sample_num = int(sample_num or getattr(opt, 'sample_num', 64))
        self.device = device
        self.net_geometry_decoder = GeometryMlpDecoder(pos_freqs=10, hidden_size=256, num_layers=10)
        self.gauge_transform = GaugeTransform(self.primitive_type)
        self.net_texture = TextureMlpDecoder(2 if self.primitive_type == 'square' else 3)
        self._handle = None
        self._key = None
        self.to(device)

    def layers(self):
        """The 29 Linear layers in the order of ngf_uv_desc."""
        g = [m for m in self.net_geometry_decoder.block if isinstance(m, nn.Linear)]
        e = self.gauge_transform.encoder
        ga = [e.linear1, e.linear2, e.linear_list[0], e.linear_list[1], e.last_linear]
        t1 = [m for m in self.net_texture.block1 if isinstance(m, nn.Linear)]
        t2 = [m for m in self.net_texture.block2 if isinstance(m, nn.Linear)]
        out = g + ga + t1 + [self.net_texture.color1] + t2
        assert len(out) == _lib.UV_LAYERS
        return out

    def load_params(self, params: dict):
        sd = {k: torch.as_tensor(v) for k, v in params.items()}
        self.load_state_dict(sd, strict=False)
        self.to(self.device)

    # --- texture editing (decoder.py:52-58, 79-121): net_texture.cubemap_ / cubemap_mode_ -------------------------------
    def set_target_texture(self, cubemap, mode=0):
        """``net_texture.cubemap_ = cubemap; net_texture.cubemap_mode_ = mode`` of the reference: a [6,R,R,C] cube map
        (sphere) or an [H,W,C] image (square), values in [0,1], C = 3 or 4 (what load_cube_from_single_texture /
        load_square return, util.py:240-275); ``None`` switches editing off."""
        self.net_texture.cubemap_ = None if cubemap is None else torch.as_tensor(cubemap, dtype=torch.float32)
        self.net_texture.cubemap_mode_ = int(mode)
        if self._handle is not None:
            self._push_texture()

    def _push_texture(self):
        t = getattr(self.net_texture, "cubemap_", None)
        dev = torch.device(self.device)
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            if t is None:
                _lib.check(_lib.lib().ngf_uv_set_texture(self._handle, None, 0, 0, 0, 0, 0, st))
                return
            t = t.to(dev, torch.float32).contiguous()
            faces = 6 if t.dim() == 4 else 1
            H, W, Cn = (t.shape[1], t.shape[2], t.shape[3]) if t.dim() == 4 else tuple(t.shape)
            _lib.check(_lib.lib().ngf_uv_set_texture(self._handle, t.data_ptr(), faces, int(H), int(W), int(Cn), int(self.net_texture.cubemap_mode_), st))
            torch.cuda.current_stream().synchronize()        # the library copied from `t`; it may be freed now

    @torch.no_grad()
    def texture_edit(self, uv, original_color):
        """The edit stage alone (decoder.py:95-121): uv [n,3] (gauge output; z ignored for square models), original_color
        = color1 + color2 [n,3] -> [n,3]."""
        dev = torch.device(self.device)
        uv = uv.to(dev, torch.float32).contiguous()
        oc = original_color.to(dev, torch.float32).contiguous()
        out = torch.empty_like(oc)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ngf_uv_texture_edit(self.handle(), uv.data_ptr(), oc.data_ptr(), uv.shape[0], out.data_ptr(),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def release(self):
        if self._handle is not None:
            _lib.lib().ngf_uv_destroy(self._handle)
            self._handle, self._key = None, None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def handle(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._handle is not None and key == self._key:
            return self._handle
        dev = torch.device(self.device)
        if dev.type != 'cuda':
            raise RuntimeError("ngf_amd NeuTex renders on the GPU only (device='cuda'); there is no CPU path")
        d = _lib.UvDesc()
        d.sphere = int(self.primitive_type != 'square')
        d.flags = _lib.UV_F_SPLIT_BF16 if self.split_bf16 else 0
        keep = []
        for i, lin in enumerate(self.layers()):
            w = lin.weight.detach().to(dev, torch.float32).contiguous()
            b = lin.bias.detach().to(dev, torch.float32).contiguous()
            keep += [w, b]
            d.w[i], d.b[i] = w.data_ptr(), b.data_ptr()
        out = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ngf_uv_create(C.byref(d), C.byref(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self.release()
        self._handle, self._key = out, key
        self._push_texture()
        return out

    @torch.no_grad()
    def forward(self, camera_position=None, ray_direction=None, background_color=None, jitter_u=None, debug=False,
                collect_stats=False):
        """model.py:27: camera_position [N,3], ray_direction [N,R,3] (normalised), background_color [N,3] or None."""
        dev = torch.device(self.device)
        N, R = ray_direction.shape[0], ray_direction.shape[1]
        S = self.sample_num
        rd = ray_direction.to(dev, torch.float32).contiguous()
        if jitter_u is None:
            jitter_u = torch.rand((N, R, S), device=dev)
        U = jitter_u.to(dev, torch.float32).contiguous()
        color = torch.empty((N, R, 3), device=dev)
        trans = torch.empty((N, R), device=dev)
        dbg_s = torch.zeros((N, R, S), device=dev) if debug else None
        dbg_c = torch.zeros((N, R, S, 3), device=dev) if debug else None
        stats = torch.zeros(16, dtype=torch.int64, device=dev) if collect_stats else None
        h = self.handle()
        # camera positions / backgrounds travel as device tensors ([N,3] in HBM): no `.cpu()`, so a chunked caller (the reference
        # renders 1024 rays per call, UV-Mapping/test.py:108-114) never synchronises, and the N cameras of a batch are ONE launch
        cam = camera_position.detach().to(dev, torch.float32).contiguous()
        bg = None if background_color is None else background_color.detach().to(dev, torch.float32).contiguous()
        if tuple(cam.shape) != (N, 3) or (bg is not None and tuple(bg.shape) != (N, 3)):
            raise ValueError(f"camera_position / background_color must be [N,3] with N = {N}")
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.lib().ngf_uv_render_batch(
                h, cam.data_ptr(), rd.data_ptr(), None if bg is None else bg.data_ptr(), U.data_ptr(), N, R, S, color.data_ptr(), trans.data_ptr(),
                None if dbg_s is None else dbg_s.data_ptr(), None if dbg_c is None else dbg_c.data_ptr(),
                None if stats is None else stats.data_ptr(), st))
        out = {"color": color, "transmittance": trans}
        if collect_stats:
            self.last_stats = stats
        if debug:
            out["sigma"], out["point_color"] = dbg_s, dbg_c
        return out
