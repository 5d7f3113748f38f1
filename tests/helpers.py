"""Shared test plumbing: rebuild a case (parameters, geometry, oracle) from a golden fixture."""
import os

import numpy as np

import ngf_amd  # noqa: F401
from ngf_amd import geometry, synth
from ngf_amd.cases import big_case, field_for_case  # noqa: F401  (re-exported: the bench builds its fields from the package)
from oracle.oracle import OracleField

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    model = str(g["model"])
    plane_hw = tuple(tuple(int(v) for v in hw) for hw in g["plane_hw"])
    if model == "triplane":
        params = synth.triplane_params(int(g["seed"]), plane_hw, tuple(int(v) for v in g["gauge_hw"]),
                                       preset=str(g["preset"]), gauge_std=float(g["gauge_std"]))
    else:
        params = synth.infoinv_params(int(g["seed"]), plane_hw, preset=str(g["preset"]))
    # the regenerated parameters must be the ones the reference saw
    for k, v in params.items():
        v64 = v.astype(np.float64).reshape(-1)
        chk = np.array([v64.sum(), np.abs(v64).sum(), v64[:: max(1, v64.size // 7)][:7].sum()])
        assert np.array_equal(chk, g["chk." + k]), f"synth regenerated different parameters for {k}"
    step = geometry.step_size(g["aabb"], g["grid"], float(g["step_ratio"]))
    if "S" not in g:
        g["S"] = np.array(0)
    mask = None
    if "mask_bits" in g:
        mask = (g["mask_bits"], tuple(int(v) for v in g["mask_dhw"]), g["mask_aabb"])
    return g, params, step, mask


def load_train_case(name):
    """Training fixture (make_golden.py capture_train): (arrays, regenerated initial parameters)."""
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    plane_hw = tuple(tuple(int(v) for v in hw) for hw in g["plane_hw"])
    params = synth.triplane_params(int(g["seed"]), plane_hw, tuple(int(v) for v in g["gauge_hw"]), preset=str(g["preset"]),
                                   gauge_std=float(g["gauge_std"]))
    for k, v in params.items():
        v64 = v.astype(np.float64).reshape(-1)
        chk = np.array([v64.sum(), np.abs(v64).sum(), v64[:: max(1, v64.size // 7)][:7].sum()])
        assert np.array_equal(chk, g["chk." + k]), f"synth regenerated different parameters for {k}"
    g.setdefault("step_ratio", np.float32(0.5))
    return g, params


def oracle_for_case(g, params, step, mask):
    return OracleField(params, g["aabb"], step, near_far=g["near_far"], distance_scale=float(g["distance_scale"]),
                       rayMarch_weight_thres=float(g["thr"]), model=str(g["model"]),
                       gauge_on=bool(int(g["gauge_on"])) if "gauge_on" in g else True,
                       infoinv=bool(int(g["infoinv"])) if "infoinv" in g else True, alpha_mask=mask)


def max_rel(a, b, atol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + atol))) if a.size else 0.0          # rel with atol 1e-6


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 200.0 if mse == 0 else -10.0 * np.log10(mse)


def load_uv_case(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    params = synth.uvmapping_params(int(g["seed"]), str(g["primitive_type"]))
    for k, v in params.items():
        v64 = v.astype(np.float64).reshape(-1)
        chk = np.array([v64.sum(), np.abs(v64).sum(), v64[:: max(1, v64.size // 7)][:7].sum()])
        assert np.array_equal(chk, g["chk." + k]), f"synth regenerated different parameters for {k}"
    return g, params
