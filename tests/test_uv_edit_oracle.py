"""CPU: the texture-editing oracle (oracle/uvedit.py) against outputs of the reference TextureMlpDecoder with cubemap_ set
(tests/golden/uv_edit.npz: sphere + square, the five cubemap_mode_ branches, ties between cube faces, border clamping)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import uvedit  # noqa: E402

G = dict(np.load(os.path.join(ROOT, "tests", "golden", "uv_edit.npz")))


@pytest.mark.parametrize("prim", ["sphere", "square"])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_edit_matches_reference(prim, mode):
    got = uvedit.texture_edit(G[f"{prim}.tex"], mode, prim == "sphere", G[f"{prim}.uv"], G[f"{prim}.orig"])
    want = G[f"{prim}.mode{mode}"][:, :3]
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    # mode 2 multiplies by 1 / texel: the rounding of the 4-tap sum is amplified
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5 if mode == 2 else 2e-6, atol=2e-7)


def test_plain_branch_is_clamped_sum():
    for prim in ("sphere", "square"):
        np.testing.assert_array_equal(G[f"{prim}.plain"], np.maximum(G[f"{prim}.plain_orig"], 0))
