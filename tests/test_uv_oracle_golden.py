"""UV-Mapping (NeuTex) colour path: the C restatement against outputs of the reference's own sub-modules
(GeometryMlpDecoder, GaugeTransform, TextureMlpDecoder, cube_ray_generation, ray_march, simple_tone_map)
composed as NeuTex.forward composes them (UV-Mapping/model/model.py:30-50)."""
import numpy as np
import pytest

from helpers import load_uv_case
from oracle.oracle import OracleUV


@pytest.mark.parametrize("name", ["uv_sphere", "uv_square"])
def test_uv_oracle_matches_reference(name):
    g, params = load_uv_case(name)
    orc = OracleUV(params, str(g["primitive_type"]))
    color, trans, dbg = orc.render(g["campos"], g["raydir"], g["U"], bg=g["bg"], debug=True)
    assert np.array_equal(dbg["valid"][:8].astype(bool), g["i_valid"].astype(bool))
    np.testing.assert_allclose(dbg["sigma"][:8], g["i_sigma"], rtol=2e-4, atol=1e-6)
    ud = 3 if str(g["primitive_type"]) == "sphere" else 2
    np.testing.assert_allclose(dbg["uv"][:8, :, :ud], g["i_uv"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(dbg["col"][:8], g["i_col"], rtol=2e-4, atol=3e-4)   # PE10(uv) amplifies 1e-6 uv differences 512x
    assert np.abs(trans - g["transmittance"]).max() <= 2e-6
    assert np.abs(color - g["color"]).max() <= 5e-6
