"""ctypes front-end of the C restatement (oracle/ngf_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Takes the reference's own parameter layout (a dict keyed by ``state_dict`` names, numpy float32,
NCHW planes) and host ray arrays; returns numpy arrays.  See ngf_oracle.c for what is restated
and how it is pinned (tests/golden/*.npz, captured from the reference import).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODEL_TRIPLANE, MODEL_INFOINV = 0, 1


class _Model(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("gauge_on", C.c_int32), ("infoinv", C.c_int32),
        ("dens_dim", C.c_int32), ("app_feat", C.c_int32),
        ("aabb", C.c_float * 6), ("near_", C.c_float), ("far_", C.c_float),
        ("step", C.c_float), ("dscale", C.c_float), ("thr", C.c_float),
        ("plane", C.c_void_p * 3), ("plane_h", C.c_int32 * 3), ("plane_w", C.c_int32 * 3),
        ("gauge", C.c_void_p * 3), ("gauge_h", C.c_int32 * 3), ("gauge_w", C.c_int32 * 3),
        ("dens_w1", C.c_void_p), ("dens_b1", C.c_void_p), ("dens_w2", C.c_void_p),
        ("dens_b2", C.c_void_p), ("dens_w3", C.c_void_p), ("dens_b3", C.c_void_p),
        ("basis", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
        ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
        ("mask_bits", C.c_void_p), ("mask_d", C.c_int32), ("mask_h", C.c_int32), ("mask_w", C.c_int32),
        ("mask_aabb", C.c_float * 6),
    ]


class _Debug(C.Structure):
    _fields_ = [("n_rays", C.c_int64), ("tmin", C.c_void_p), ("z", C.c_void_p), ("valid", C.c_void_p),
                ("sigma", C.c_void_p), ("alpha", C.c_void_p), ("weight", C.c_void_p),
                ("active", C.c_void_p), ("rgb", C.c_void_p), ("coords", C.c_void_p)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libngf_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("ngf_oracle.c", "ngf_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libngf_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ngf_oracle_render.restype = C.c_int
        _LIB.ngf_oracle_render.argtypes = [C.POINTER(_Model), C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Debug), C.c_int32]
    return _LIB


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleField:
    """Holds the reference-layout parameters and renders rays with the C restatement."""

    def __init__(self, params: dict, aabb, step: float, near_far=(2.0, 6.0), distance_scale=25.0,
                 rayMarch_weight_thres=1e-4, model="triplane", gauge_on=True, infoinv=True,
                 alpha_mask=None):
        """alpha_mask = (packbits uint8 array, (D,H,W), mask_aabb[2,3]) or None."""
        self._keep = {k: _f32(v) for k, v in params.items()}
        p = self._keep
        m = _Model()
        self.is_infoinv = model == "infoinv"
        m.model = MODEL_INFOINV if self.is_infoinv else MODEL_TRIPLANE
        m.gauge_on = int(bool(gauge_on))
        m.infoinv = int(bool(infoinv))
        m.dens_dim = 24 if self.is_infoinv else 16
        C_total = p["plane_xy"].shape[1]
        m.app_feat = 3 * (C_total - m.dens_dim)
        m.aabb = (C.c_float * 6)(*np.asarray(aabb, np.float32).reshape(-1))
        m.near_, m.far_ = float(near_far[0]), float(near_far[1])
        m.step = float(np.float32(step))
        m.dscale = float(distance_scale)
        m.thr = float(np.float32(rayMarch_weight_thres))
        for k, name in enumerate(("xy", "yz", "xz")):
            pl = p[f"plane_{name}"]
            m.plane[k] = pl.ctypes.data
            m.plane_h[k], m.plane_w[k] = pl.shape[2], pl.shape[3]
            if not self.is_infoinv:
                g = p[f"gauge_{name}"]
                m.gauge[k] = g.ctypes.data
                m.gauge_h[k], m.gauge_w[k] = g.shape[2], g.shape[3]
        if self.is_infoinv:
            m.dens_w1, m.dens_b1 = _ptr(p["density_decoder.mlp.0.weight"]), _ptr(p["density_decoder.mlp.0.bias"])
            m.dens_w2, m.dens_b2 = _ptr(p["density_decoder.mlp.2.weight"]), _ptr(p["density_decoder.mlp.2.bias"])
            m.dens_w3, m.dens_b3 = _ptr(p["density_decoder.mlp.4.weight"]), _ptr(p["density_decoder.mlp.4.bias"])
        else:
            m.dens_w1, m.dens_b1 = _ptr(p["density_decoder.weight"]), _ptr(p["density_decoder.bias"])
        m.basis = _ptr(p["rgb_decoder.basis.weight"])
        m.w1, m.b1 = _ptr(p["rgb_decoder.mlp.0.weight"]), _ptr(p["rgb_decoder.mlp.0.bias"])
        m.w2, m.b2 = _ptr(p["rgb_decoder.mlp.2.weight"]), _ptr(p["rgb_decoder.mlp.2.bias"])
        m.w3, m.b3 = _ptr(p["rgb_decoder.mlp.4.weight"]), _ptr(p["rgb_decoder.mlp.4.bias"])
        if alpha_mask is not None:
            bits, dhw, maabb = alpha_mask
            self._mask = np.ascontiguousarray(np.asarray(bits, np.uint8))
            m.mask_bits = self._mask.ctypes.data
            m.mask_d, m.mask_h, m.mask_w = (int(v) for v in dhw)
            m.mask_aabb = (C.c_float * 6)(*np.asarray(maabb, np.float32).reshape(-1))
        self._m = m

    def render(self, rays, N_samples: int, white_bg=True, jitter=None, debug_rays=0, threads=None):
        rays = _f32(rays)
        n = rays.shape[0]
        S = int(N_samples)
        rgb = np.empty((n, 3), np.float32)
        depth = np.empty((n,), np.float32)
        jit = None if jitter is None else _f32(jitter)
        dbg = None
        out = {}
        if debug_rays:
            k = min(int(debug_rays), n)
            out = {"tmin": np.zeros((k,), np.float32), "z": np.zeros((k, S), np.float32),
                   "valid": np.zeros((k, S), np.uint8), "sigma": np.zeros((k, S), np.float32),
                   "alpha": np.zeros((k, S), np.float32), "weight": np.zeros((k, S), np.float32),
                   "active": np.zeros((k, S), np.uint8), "rgb": np.zeros((k, S, 3), np.float32),
                   "coords": np.zeros((k, S, 6), np.float32)}
            dbg = _Debug(k, *[out[f].ctypes.data for f in
                              ("tmin", "z", "valid", "sigma", "alpha", "weight", "active", "rgb", "coords")])
        if threads is None:
            threads = os.cpu_count() or 1
        rc = lib().ngf_oracle_render(C.byref(self._m), _ptr(rays), n, S, int(bool(white_bg)),
                                     None if jit is None else _ptr(jit), _ptr(rgb), _ptr(depth),
                                     None if dbg is None else C.byref(dbg), int(threads))
        if rc != 0:
            raise RuntimeError(f"ngf_oracle_render failed: {rc}")
        if debug_rays:
            return rgb, depth, out
        return rgb, depth

    def color_at(self, coords, dirs):
        coords, dirs = _f32(coords), _f32(dirs)
        out = np.empty((coords.shape[0], 3), np.float32)
        lib().ngf_oracle_color_at(C.byref(self._m), _ptr(coords), _ptr(dirs), C.c_int64(coords.shape[0]), _ptr(out))
        return out

    def density_at(self, coords):
        coords = _f32(coords)
        out = np.empty((coords.shape[0],), np.float32)
        lib().ngf_oracle_density_at(C.byref(self._m), _ptr(coords), C.c_int64(coords.shape[0]), _ptr(out))
        return out


# ---- UV-Mapping (NeuTex) colour path -------------------------------------------------------------------------------
class _Linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("in_f", C.c_int32), ("out_f", C.c_int32), ("act", C.c_int32)]


class _UvModel(C.Structure):
    _fields_ = [("sphere", C.c_int32), ("geo", _Linear * 12), ("gauge", _Linear * 5), ("tex1", _Linear * 6),
                ("color1", _Linear), ("tex2", _Linear * 5)]


def uv_layer_table(primitive_type="sphere"):
    """(state_dict prefix, activation) of every Linear on the colour path, in evaluation order, per sub-network.
    act: 0 none, 1 ReLU, 2 LeakyReLU(0.2)."""
    geo = [(f"net_geometry_decoder.block.{2 * i}", 1) for i in range(11)] + [("net_geometry_decoder.block.22", 0)]
    gauge = [("gauge_transform.encoder.linear1", 1), ("gauge_transform.encoder.linear2", 1),
             ("gauge_transform.encoder.linear_list.0", 1), ("gauge_transform.encoder.linear_list.1", 1),
             ("gauge_transform.encoder.last_linear", 0)]
    tex1 = [(f"net_texture.block1.{2 * i}", 2) for i in range(6)]
    tex2 = [(f"net_texture.block2.{2 * i}", 2) for i in range(4)] + [("net_texture.block2.8", 0)]
    return {"geo": geo, "gauge": gauge, "tex1": tex1, "color1": [("net_texture.color1", 0)], "tex2": tex2}


class OracleUV:
    def __init__(self, params: dict, primitive_type="sphere"):
        self._keep = {k: _f32(v) for k, v in params.items()}
        m = _UvModel()
        m.sphere = int(primitive_type == "sphere")
        tab = uv_layer_table(primitive_type)

        def fill(dst, name, act):
            w, b = self._keep[name + ".weight"], self._keep[name + ".bias"]
            dst.w, dst.b, dst.in_f, dst.out_f, dst.act = w.ctypes.data, b.ctypes.data, w.shape[1], w.shape[0], act

        for key, arr in (("geo", m.geo), ("gauge", m.gauge), ("tex1", m.tex1), ("tex2", m.tex2)):
            for i, (name, act) in enumerate(tab[key]):
                fill(arr[i], name, act)
        fill(m.color1, "net_texture.color1", 0)
        self._m = m
        L = lib()
        L.ngf_oracle_uv_render.restype = C.c_int

    def render(self, campos, raydir, U, bg=None, debug=False, threads=None):
        campos, raydir, U = _f32(campos).reshape(3), _f32(raydir), _f32(U)
        R, S = U.shape
        color = np.empty((R, 3), np.float32)
        trans = np.empty((R,), np.float32)
        bgp = None if bg is None else _f32(bg).reshape(3)
        dbg = {}
        if debug:
            dbg = {"sigma": np.zeros((R, S), np.float32), "uv": np.zeros((R, S, 3), np.float32),
                   "col": np.zeros((R, S, 3), np.float32), "valid": np.zeros((R, S), np.uint8)}
        if threads is None:
            threads = os.cpu_count() or 1
        rc = lib().ngf_oracle_uv_render(C.byref(self._m), _ptr(campos), _ptr(raydir), None if bgp is None else _ptr(bgp),
                                        _ptr(U), C.c_int64(R), C.c_int32(S), _ptr(color), _ptr(trans),
                                        _ptr(dbg["sigma"]) if debug else None, _ptr(dbg["uv"]) if debug else None,
                                        _ptr(dbg["col"]) if debug else None, _ptr(dbg["valid"]) if debug else None,
                                        C.c_int32(int(threads)))
        if rc != 0:
            raise RuntimeError(f"ngf_oracle_uv_render failed: {rc}")
        return (color, trans, dbg) if debug else (color, trans)
