"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/ngf.h declares (no compute without a GPU), the ctypes struct matches the C layout, and the
host-side mirror keeps the reference's names (state_dict keys, checkpoint dict, signatures)."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

import ngf_amd  # noqa: F401
from ngf_amd import _lib, infoinv, triplane

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ngf.h")).read()
    declared = set(re.findall(r"\b(ngf_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.ngf_abi_version() == 5
    import ctypes as C
    assert L.ngf_sizeof_field_desc() == C.sizeof(_lib.FieldDesc)
    # the experiment library (product kernels + the experiment kernels; tests and profiles/ only) exports the same ABI
    with _lib.library("exp") as X:
        assert X is not L and all(hasattr(X, s) for s in declared) and X.ngf_abi_version() == 5
    assert _lib.lib() is L


def test_error_path_without_gpu():
    import ctypes as C
    L = _lib.lib()
    out = C.c_void_p()
    assert L.ngf_field_create(None, C.byref(out), None) == 1
    assert b"null" in L.ngf_last_error()
    assert L.ngf_field_render(None, None, 0, 0, 0, 0, None, None, None, None, None) == 1


def test_state_dict_names_and_signatures_match_reference():
    aabb = torch.tensor([[-1.5] * 3, [1.5] * 3])
    f = triplane.TriPlane(aabb, [256] * 3, "cpu", near_far=[2.0, 6.0], step_ratio=0.5)
    names = {k: tuple(v.shape) for k, v in f.state_dict().items()}
    assert names == {  # SURVEY.md Appendix B, measured on the reference via named_parameters()
        "plane_xy": (1, 64, 256, 256), "plane_yz": (1, 64, 256, 256), "plane_xz": (1, 64, 256, 256),
        "gauge_xy": (1, 2, 256, 256), "gauge_yz": (1, 2, 256, 256), "gauge_xz": (1, 2, 256, 256),
        "rgb_decoder.basis.weight": (144, 144), "rgb_decoder.mlp.0.weight": (64, 159), "rgb_decoder.mlp.0.bias": (64,),
        "rgb_decoder.mlp.2.weight": (64, 64), "rgb_decoder.mlp.2.bias": (64,), "rgb_decoder.mlp.4.weight": (3, 64),
        "rgb_decoder.mlp.4.bias": (3,), "density_decoder.weight": (1, 48), "density_decoder.bias": (1,)}
    assert f.nSamples == 884 and abs(float(f.stepSize) - 0.0058823532) < 1e-9
    sig = inspect.signature(f.forward)
    assert list(sig.parameters)[:5] == ["rays_chunk", "white_bg", "is_train", "N_samples", "iteration"]
    assert [sig.parameters[k].default for k in ("white_bg", "is_train", "N_samples", "iteration")] == [True, False, -1, 0]
    g = infoinv.TriPlane(aabb, [256] * 3, "cpu", step_ratio=0.5)
    gn = {k: tuple(v.shape) for k, v in g.state_dict().items()}
    assert gn["plane_xy"] == (1, 96, 256, 256) and gn["density_decoder.mlp.0.weight"] == (32, 72)
    assert gn["rgb_decoder.basis.weight"] == (216, 216) and gn["rgb_decoder.mlp.0.weight"] == (64, 231)
    assert list(inspect.signature(g.forward).parameters)[4] == "infoinv"
    rsig = inspect.signature(triplane.renderer)
    assert list(rsig.parameters)[:7] == ["rays", "field", "chunk", "N_samples", "white_bg", "is_train", "device"]


def test_checkpoint_roundtrip_format(tmp_path):
    aabb = torch.tensor([[-1.5] * 3, [1.5] * 3])
    f = triplane.TriPlane(aabb, [32, 30, 28], "cpu", step_ratio=0.5)
    f.plane_xy = torch.nn.Parameter(torch.randn(1, 64, 30, 32))          # up-sampled / shrunk planes
    vol = (torch.rand(9, 10, 11) > 0.5).float()
    f.alphaMask = triplane.AlphaGridMask("cpu", aabb * 0.9, vol)
    p = str(tmp_path / "ck.th")
    f.save(p)
    ck = torch.load(p, weights_only=False)
    assert set(ck) == {"kwargs", "state_dict", "alphaMask.shape", "alphaMask.mask", "alphaMask.aabb"}
    assert set(ck["kwargs"]) == {"aabb", "gridSize", "alphaMask_thres", "distance_scale", "rayMarch_weight_thres",
                                 "near_far", "step_ratio"}
    assert ck["alphaMask.mask"].dtype == np.uint8 and tuple(ck["alphaMask.shape"]) == (1, 1, 9, 10, 11)
    kw = dict(ck["kwargs"])
    kw["device"] = "cpu"
    f2 = triplane.TriPlane(**kw)
    f2.load(ck)                                                            # sizes planes from the state_dict
    assert f2.plane_xy.shape == (1, 64, 30, 32) and torch.equal(f2.plane_xy, f.plane_xy)
    assert torch.equal(f2.alphaMask.alpha_volume, f.alphaMask.alpha_volume)


def test_cpu_device_fails_loudly():
    aabb = torch.tensor([[-1.5] * 3, [1.5] * 3])
    f = triplane.TriPlane(aabb, [16] * 3, "cpu", step_ratio=0.5)
    with pytest.raises(RuntimeError, match="GPU only"):
        f(torch.zeros(4, 6), N_samples=4)


def test_loads_a_checkpoint_written_by_the_reference():
    """tests/golden/ref_ckpt_triplane.th was written by the reference's Base.save (make_golden.py capture_ref_checkpoint); the
    drop-in builds from its kwargs and loads it exactly like TriPlane/main.py:34-38 does."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = torch.load(os.path.join(root, "tests", "golden", "ref_ckpt_triplane.th"), map_location="cpu", weights_only=False)
    kw = dict(ck["kwargs"])
    kw.update({"device": "cpu"})
    f = triplane.TriPlane(**kw)
    f.load(ck)
    g = np.load(os.path.join(root, "tests", "golden", "ref_ckpt_triplane.npz"))
    from ngf_amd import synth
    params = synth.triplane_params(int(g["seed"]), tuple(tuple(int(v) for v in hw) for hw in g["plane_hw"]), tuple(int(v) for v in g["gauge_hw"]),
                                   preset="R1", gauge_std=0.04)
    sd = f.state_dict()
    assert set(sd) == set(params)
    for k, v in params.items():
        assert np.array_equal(sd[k].numpy(), v), k
    vol, _ = synth.alpha_mask_bits(int(g["seed"]), tuple(int(v) for v in g["mask_dhw"]))
    assert np.array_equal(f.alphaMask.alpha_volume[0, 0].numpy(), vol.astype(np.float32))
    assert abs(float(f.stepSize) - 0.5 * float(np.mean((np.float32(3.0) / (np.array([12, 10, 9], np.float32) - 1))))) < 1e-6
