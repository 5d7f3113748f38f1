"""Scalar geometry of the field -- the host-side arithmetic of ``Base.init_para``
(TriPlane/models/FieldBase.py:63-74) and the two resolution helpers the drivers use
(TriPlane/utils.py:74-80).  float32 throughout, one rounding per operation, because
``stepSize`` feeds every sample position and must match the reference bit for bit.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32


def step_size(aabb, grid_size, step_ratio: float) -> np.float32:
    """stepSize = mean(aabbSize / (gridSize - 1)) * step_ratio (FieldBase.py:66-70)."""
    aabb = np.asarray(aabb, f32).reshape(2, 3)
    size = aabb[1] - aabb[0]
    units = size / (np.asarray(grid_size, np.int64) - 1).astype(f32)
    mean = ((units[0] + units[1]) + units[2]) / f32(3.0)
    return f32(mean * f32(step_ratio))


def n_samples(aabb, step: np.float32) -> int:
    """nSamples = int(aabbDiag / stepSize) + 1 (FieldBase.py:71-72)."""
    aabb = np.asarray(aabb, f32).reshape(2, 3)
    size = aabb[1] - aabb[0]
    sq = size * size
    diag = np.sqrt(f32((sq[0] + sq[1]) + sq[2]))
    return int(f32(diag / f32(step))) + 1


def N_to_reso(n_voxels, bbox):
    """Grid resolution for a voxel budget (TriPlane/utils.py:74-77)."""
    bbox = np.asarray(bbox, f32).reshape(2, 3)
    xyz = bbox[1] - bbox[0]
    # the reference computes this with torch float32 tensors: (xyz_max - xyz_min).prod() / n_voxels
    vol = f32(f32(xyz[0] * xyz[1]) * xyz[2])
    voxel = f32(np.power(f32(vol / f32(n_voxels)), f32(1.0 / 3.0)))
    return [int(v) for v in (xyz / voxel).astype(np.int64)]


def cal_n_samples(reso, step_ratio=0.5):
    """TriPlane/utils.py:79-80."""
    return int(np.linalg.norm(reso) / step_ratio)
