"""Host-side check of the RULE behind the march's empty-space skipping (csrc/ngf_device.hpp mask_clear_around, csrc/ngf_field.hip mask_coarse_*_kernel,
csrc/ngf_render.hpp: the empty-iteration branch).  The kernels themselves are compared bit for bit with the unskipped march on the GPU
(tests/test_gpu_parity.py::test_empty_space_skipping_is_bit_identical); here the two claims they rest on are restated in numpy (fp32 like the device
code) and tried against the oracle's grid_sample on small random volumes:
  (a) block images (blocks of B = 8 and of B = 4 cells): `clear` at the block of a sample's cell  =>  every sample whose cell index differs by at most B per axis samples the mask as 0
      (cells outside the volume included -- zeros padding);
  (b) step rule: with r = cells per step of a ray (|d_k| step inv_k / 2 (size_k - 1), largest axis) and m = floor((B - 1.1) / r), the samples of the steps
      within m of a step have cell indices within B of that step's.
No GPU, no product code: the oracle is the checker (test infrastructure)."""
import ctypes as C

import numpy as np
import pytest

import ngf_amd  # noqa: F401
from ngf_amd import synth
from oracle import oracle as O

F = np.float32


def _cell_index(p, a0, inv, dhw):
    """floor of the grid_sample index per axis (x, y, z), as mask_occupied / mask_clear_around compute it (fp32)."""
    d, h, w = dhw
    q = ((p - a0) * inv - F(1.0)).astype(F)
    sz = np.array([w - 1, h - 1, d - 1], F)
    return np.floor(((q + F(1.0)) / F(2.0)) * sz).astype(F)


def _images(vol, log=3):
    """numpy restatement of mask_cells_kernel (any corner of the cell set, as a bool) and of the two block passes (margin 16 cells, blocks of B = 2^log cells)."""
    B = 1 << log
    d, h, w = vol.shape
    pad = np.zeros((d + 2, h + 2, w + 2), bool)
    pad[1:-1, 1:-1, 1:-1] = vol
    cells = np.zeros((d + 1, h + 1, w + 1), bool)          # cell (z1, y1, x1) = base corner (z1 - 1, ...) in volume coordinates
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                cells |= pad[dz:dz + d + 1, dy:dy + h + 1, dx:dx + w + 1]
    cD, cH, cW = ((d + 32) >> log) + 1, ((h + 32) >> log) + 1, ((w + 32) >> log) + 1
    ext = np.zeros((cD * B, cH * B, cW * B), bool)          # index = cell index + 16
    ext[16:16 + d + 1, 16:16 + h + 1, 16:16 + w + 1] = cells
    anyb = ext.reshape(cD, B, cH, B, cW, B).any(axis=(1, 3, 5))
    pb = np.zeros((cD + 2, cH + 2, cW + 2), bool)
    pb[1:-1, 1:-1, 1:-1] = anyb
    near = np.zeros_like(anyb)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                near |= pb[dz:dz + cD, dy:dy + cH, dx:dx + cW]
    return ~near


def _clear_around(clear, fl, dhw, log=3):
    d, h, w = dhw
    X = np.minimum(np.maximum(fl[:, 0] + F(17.0), F(0.0)), F(w + 32)).astype(np.int64) >> log
    Y = np.minimum(np.maximum(fl[:, 1] + F(17.0), F(0.0)), F(h + 32)).astype(np.int64) >> log
    Z = np.minimum(np.maximum(fl[:, 2] + F(17.0), F(0.0)), F(d + 32)).astype(np.int64) >> log
    return clear[Z, Y, X]


def _sample(bits, dhw, qn):
    out = np.zeros((qn.shape[0],), F)
    qn = np.ascontiguousarray(qn.astype(F))
    O.lib().ngf_oracle_mask_sample(bits.ctypes.data_as(C.c_void_p), dhw[0], dhw[1], dhw[2], qn.ctypes.data_as(C.c_void_p), C.c_int64(qn.shape[0]),
                                   out.ctypes.data_as(C.c_void_p))
    return out


@pytest.mark.parametrize("log", [3, 2])
@pytest.mark.parametrize("dhw,kind", [((40, 33, 47), "sparse"), ((24, 24, 24), "ball"), ((20, 31, 18), "blobby"), ((17, 17, 17), "full")])
def test_clear_block_means_nothing_occupied_within_a_block_size(dhw, kind, log):
    B = 1 << log
    rng = np.random.default_rng(7)
    d, h, w = dhw
    if kind == "sparse":
        vol = np.zeros(dhw, bool)
        vol[3:6, 20:25, 30:40] = True
        vol[30:38, 2:6, 1:4] = True
        vol[d - 1, h - 1, w - 1] = True                        # a corner voxel: the border cells with zero-padded corners
    elif kind == "ball":
        zz, yy, xx = np.meshgrid(*(np.linspace(-1, 1, n) for n in dhw), indexing="ij")
        vol = xx ** 2 + yy ** 2 + zz ** 2 < 0.3 ** 2
    elif kind == "blobby":
        vol = synth.alpha_mask_bits(5, dhw, keep=0.15)[0]
    else:
        vol = np.ones(dhw, bool)
    bits = np.packbits(vol.reshape(-1))
    a0, a1 = np.array([-1.0, -0.7, -1.2], F), np.array([1.0, 0.9, 1.1], F)
    inv = (F(1.0) / (a1 - a0) * F(2.0)).astype(F)
    clear = _images(vol, log)
    ext = (a1 - a0)
    p = (rng.uniform(-0.8, 1.8, (6000, 3)) * ext + a0).astype(F)          # in and well around the mask's box
    fl = _cell_index(p, a0, inv, dhw)
    ok = _clear_around(clear, fl, dhw, log)
    assert ok.any() and not ok.all()
    # neighbours: cell index within 8 per axis of a certified sample's, placed anywhere inside that cell
    cellsz = ext / np.array([w - 1, h - 1, d - 1], F)
    worst = 0
    for _ in range(12):
        off = rng.integers(-B, B + 1, (p.shape[0], 3)).astype(F)
        frac = rng.uniform(0.02, 0.98, (p.shape[0], 3)).astype(F)
        p2 = (a0 + ((fl + off + frac) * cellsz)).astype(F)
        fl2 = _cell_index(p2, a0, inv, dhw)
        inside = np.all(np.abs(fl2 - fl) <= B, axis=1) & ok          # (fp32 placement can land one cell off: keep the ones that are within B)
        val = _sample(bits, dhw, ((p2 - a0) * inv - F(1.0)).astype(F))
        assert not (val[inside] > 0).any(), f"{int((val[inside] > 0).sum())} occupied samples within {B} cells of a certified one"
        worst += int(inside.sum())
    assert worst > 1000
    if kind == "full":          # far outside a fully occupied volume IS certified (two blocks of margin), right next to it is not
        far = np.array([[a1[0] + 3 * ext[0], 0, 0], [a0[0] - 3 * ext[0], 0.1, 0.2], [0, a1[1] + ext[1], 0]], F)
        assert _clear_around(clear, _cell_index(far, a0, inv, dhw), dhw, log).all()
        near = np.array([[a1[0] + 2 * cellsz[0], 0, 0]], F)
        assert not _clear_around(clear, _cell_index(near, a0, inv, dhw), dhw, log).any()


@pytest.mark.parametrize("B", [8, 4])
def test_step_rule_keeps_cell_indices_within_a_block_size(B):
    rng = np.random.default_rng(11)
    dhw = (256, 200, 96)
    a0, a1 = np.array([-1.5, -1.5, -1.5], F), np.array([1.5, 1.2, 1.4], F)
    inv = (F(1.0) / (a1 - a0) * F(2.0)).astype(F)
    sz = np.array([dhw[2] - 1, dhw[1] - 1, dhw[0] - 1], F)
    step = F(0.0058823532)
    n = 4000
    o = rng.uniform(-4, 4, (n, 3)).astype(F)
    dirs = rng.normal(size=(n, 3))
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True) * rng.uniform(0.3, 1.7, (n, 1))).astype(F)      # not normalised on purpose
    tmin = rng.uniform(0, 6, n).astype(F)
    jit = rng.uniform(0, 1, n).astype(F)
    r = np.max(np.abs(dirs) * (step * F(0.5) * inv * sz), axis=1).astype(F)
    m = np.minimum((F(B) - F(1.1)) / np.maximum(r, F(1e-6)), F(2048.0)).astype(np.int64)
    s0 = rng.integers(0, 900, n)

    def cell(s):
        z = (tmin + step * (s.astype(F) + jit)).astype(F)
        p = (o + dirs * z[:, None]).astype(F)
        return _cell_index(p, a0, inv, dhw)
    c0 = cell(s0)
    for sign in (-1, 1):
        for frac in (1.0, 0.5, 0.31):
            ds = (sign * np.floor(m * frac)).astype(np.int64)
            assert np.all(np.abs(cell(s0 + ds) - c0) <= B)
    assert m.min() >= (3 if B == 8 else 1) and m.max() > (20 if B == 8 else 8)
