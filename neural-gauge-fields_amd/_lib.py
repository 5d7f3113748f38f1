"""ctypes binding of libngf_hip.so (include/ngf.h).  There is no fallback: if the HIP library is
missing or fails to load, importing a field and rendering raises -- the product path never routes
through the CPU oracle."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO_PATH = os.path.join(_CSRC, "libngf_hip.so")
SO_PATH_EXP = os.path.join(_CSRC, "libngf_hip_exp.so")      # the same + the experiment kernels (csrc/Makefile): tests / profiles scripts only
_LIB = None
_LIBS = {}                                                    # path -> loaded library

MODEL_TRIPLANE, MODEL_INFOINV = 0, 1
F_BAKE_DENSITY = 1
F_BAKE_COLOR = 2
F_NO_FOLD = 4
F_SPLIT_BF16 = 8

SYMBOLS = ["ngf_field_create", "ngf_field_destroy", "ngf_field_render", "ngf_field_render_image", "ngf_field_decode_rgb", "ngf_field_march",
           "ngf_generate_rays", "ngf_generate_rays_dtu", "ngf_last_error", "ngf_abi_version", "ngf_field_bytes", "ngf_sizeof_field_desc",
           "ngf_uv_create", "ngf_uv_destroy", "ngf_uv_render", "ngf_uv_render_batch", "ngf_field_alpha", "ngf_field_ray_filter",
           "ngf_eval_workspace_bytes", "ngf_eval_frame_u8", "ngf_eval_depth_range", "ngf_eval_depth_colormap", "ngf_eval_mse",
           "ngf_eval_ssim", "ngf_trainer_create", "ngf_trainer_destroy", "ngf_trainer_bytes", "ngf_sizeof_train_desc", "ngf_train_backward", "ngf_train_backward2",
           "ngf_train_forward", "ngf_train_backward_grad", "ngf_train_get_grad", "ngf_train_get_active", "ngf_train_overflow_count", "ngf_train_adam", "ngf_train_adam_all", "ngf_train_adam_ext", "ngf_train_get_grads", "ngf_train_params_changed", "ngf_train_debug_sections", "ngf_resize_bilinear", "ngf_uv_set_texture", "ngf_uv_texture_edit", "ngf_field_alpha_mask_build", "ngf_pack_mask_bits", "ngf_debug_set", "ngf_debug_get", "ngf_debug_dirty_lds", "ngf_debug_xcd_histogram", "ngf_debug_tile_plan", "ngf_debug_tile_order", "ngf_planes_l1", "ngf_planes_l1_backward", "ngf_pool_trim", "ngf_pool_set_limit", "ngf_pool_bytes"]


class FieldDesc(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("flags", C.c_int32), ("plane_c", C.c_int32), ("dens_dim", C.c_int32),
        ("plane", C.c_void_p * 3), ("plane_h", C.c_int32 * 3), ("plane_w", C.c_int32 * 3),
        ("gauge", C.c_void_p * 3), ("gauge_h", C.c_int32 * 3), ("gauge_w", C.c_int32 * 3),
        ("dens_w1", C.c_void_p), ("dens_b1", C.c_void_p), ("dens_w2", C.c_void_p),
        ("dens_b2", C.c_void_p), ("dens_w3", C.c_void_p), ("dens_b3", C.c_void_p),
        ("basis", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
        ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
        ("aabb", C.c_float * 6), ("near_", C.c_float), ("far_", C.c_float), ("step", C.c_float),
        ("distance_scale", C.c_float), ("weight_thres", C.c_float),
        ("mask_bits", C.c_void_p), ("mask_d", C.c_int32), ("mask_h", C.c_int32), ("mask_w", C.c_int32),
        ("mask_aabb", C.c_float * 6),
    ]


UV_LAYERS = 29


class UvDesc(C.Structure):
    _fields_ = [("sphere", C.c_int32), ("flags", C.c_int32), ("w", C.c_void_p * UV_LAYERS), ("b", C.c_void_p * UV_LAYERS)]


UV_F_SPLIT_BF16 = 1


def build(force: bool = False) -> str:
    """Compile libngf_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-j4", "-C", _CSRC, "all"] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return SO_PATH


def lib():
    """The library every call of the package goes through: libngf_hip.so, or -- inside ``with library("exp")`` -- libngf_hip_exp.so."""
    global _LIB
    if _LIB is None:
        _LIB = _load(SO_PATH)
    return _LIB


def _load(path):
    if path not in _LIBS:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is the only render path; there is no CPU fallback)")
        L = C.CDLL(path)
        L.ngf_last_error.restype = C.c_char_p
        L.ngf_abi_version.restype = C.c_int
        L.ngf_field_bytes.restype = C.c_int64
        L.ngf_field_bytes.argtypes = [C.c_void_p]
        L.ngf_field_create.argtypes = [C.POINTER(FieldDesc), C.POINTER(C.c_void_p), C.c_void_p]
        L.ngf_field_destroy.argtypes = [C.c_void_p]
        L.ngf_field_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_field_render_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_field_decode_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        L.ngf_field_march.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.ngf_generate_rays.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p]
        L.ngf_generate_rays_dtu.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p]
        L.ngf_field_alpha.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
        L.ngf_field_alpha_mask_build.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                                 C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_field_ray_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        L.ngf_uv_create.argtypes = [C.POINTER(UvDesc), C.POINTER(C.c_void_p), C.c_void_p]
        L.ngf_uv_destroy.argtypes = [C.c_void_p]
        L.ngf_uv_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_uv_render_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_uv_set_texture.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.ngf_uv_texture_edit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.ngf_pack_mask_bits.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.ngf_resize_bilinear.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.ngf_eval_workspace_bytes.restype = C.c_int64
        L.ngf_eval_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.ngf_eval_frame_u8.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.ngf_eval_depth_range.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_eval_depth_colormap.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_eval_mse.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_eval_ssim.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_double,
                                    C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_debug_set.argtypes = [C.c_char_p, C.c_int32]
        L.ngf_debug_get.argtypes = [C.c_char_p]
        L.ngf_debug_get.restype = C.c_int32
        L.ngf_debug_dirty_lds.argtypes = [C.c_void_p]
        L.ngf_debug_xcd_histogram.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ngf_debug_tile_plan.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.ngf_pool_set_limit.argtypes = [C.c_int64]
        L.ngf_pool_bytes.argtypes = [C.c_int32]
        L.ngf_pool_bytes.restype = C.c_int64
        L.ngf_planes_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_planes_l1_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngf_debug_tile_order.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        if L.ngf_abi_version() != 5 or L.ngf_sizeof_field_desc() != C.sizeof(FieldDesc):
            raise RuntimeError("libngf_hip.so ABI mismatch (version or ngf_field_desc layout)")
        _LIBS[path] = L
    return _LIBS[path]


class library:
    """``with _lib.library("exp"): ...`` -- route the package through libngf_hip_exp.so (the product kernels + the experiment kernels:
    knobs kernel / stage / profile / nstep / waves) for the duration of the block.  The two libraries are separate images with separate
    knob state: handles created inside the block must be used and released inside it.  Tests and profiles/ scripts only."""

    def __init__(self, which="exp"):
        self.path = SO_PATH_EXP if which == "exp" else SO_PATH

    def __enter__(self):
        global _LIB
        self.prev = _LIB
        _LIB = _load(self.path)
        return _LIB

    def __exit__(self, *exc):
        global _LIB
        _LIB = self.prev
        return False


E_STALE = 4          # ngf_train_backward_grad only: the ticket is not the trainer's last forward


def check(rc: int):
    if rc != 0:
        raise RuntimeError("libngf_hip: " + (lib().ngf_last_error() or b"?").decode())


class knobs:
    """Context manager over ngf_debug_set: ``with _lib.knobs(tile_w=8, split=1): ...`` -- experiment / test knobs of the launch
    code, restored to their previous values on exit.  (The library does not read environment variables.)"""

    def __init__(self, **kw):
        self.kw = kw
        self.old = {}

    def __enter__(self):
        L = lib()
        for k, v in self.kw.items():
            self.old[k] = L.ngf_debug_get(k.encode())
            check(L.ngf_debug_set(k.encode(), int(v)))
        return self

    def __exit__(self, *exc):
        L = lib()
        for k, v in self.old.items():
            L.ngf_debug_set(k.encode(), int(v))
        return False


def knobs_from_env():
    """Opt-in for the experiment scripts under profiles/: map NGF_TILE_W / NGF_SPLIT / NGF_WAVES / NGF_NSTEP / NGF_PROFILE /
    NGF_ABLATE / NGF_UV_TILES / NGF_KERNEL / NGF_STAGE of the environment to ngf_debug_set calls (unset -> library default).
    Nothing in the product path calls this."""
    global _LIB
    if any(os.environ.get("NGF_" + k.upper()) not in (None, "", "-1") for k in ("waves", "nstep", "profile", "kernel", "stage")):
        _LIB = _load(SO_PATH_EXP)          # these knobs select experiment kernels: the whole script runs on libngf_hip_exp.so
    L = lib()
    for k in ("tile_w", "split", "waves", "nstep", "profile", "ablate", "uv_tiles", "kernel", "stage", "xcd", "grid", "tail", "ord_rows", "ord_px", "train_dwg"):
        v = os.environ.get("NGF_" + k.upper())
        check(L.ngf_debug_set(k.encode(), int(v) if v not in (None, "") else -1))
