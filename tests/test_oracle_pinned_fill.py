"""oracle/fill_oracle.py (diffusion fill restatement) against golden outputs of the reference's own regionfill
(tests/golden/make_golden_fill.py), plus the Laplace-equation property every solver of this stage must satisfy."""
import os

import numpy as np
import pytest

from oracle import fill_oracle as FO
from util import GOLDEN

CASES = ["blobs", "borders", "empty", "pixels", "large"]


@pytest.mark.parametrize("name", CASES)
def test_regionfill_matches_reference_golden(name):
    g = np.load(os.path.join(GOLDEN, f"fill_{name}.npz"))
    out = FO.regionfill(g["I"], g["mask"])
    assert out.shape == g["out"].shape
    assert np.abs(out - g["out"]).max() <= 1e-9 * max(1.0, np.abs(g["out"]).max())
    m = g["mask"] != 0
    assert np.array_equal(out[~m], g["I"][~m].astype(float))
    assert FO.residual(g["out"], g["I"], g["mask"]) < 1e-9          # the reference's output solves the stencil equation
    assert FO.residual(out, g["I"], g["mask"]) < 1e-9


def test_diffusion_applies_the_mask_to_both_channels():
    rng = np.random.default_rng(0)
    flows = rng.standard_normal((3, 24, 32, 2)).astype(np.float32)
    masks = np.zeros((3, 24, 32, 1), dtype=np.uint8)
    masks[0, 5:12, 8:20] = 1
    masks[2, :, :4] = 1
    d = FO.diffusion(flows, masks)
    assert d.shape == flows.shape and d.dtype == np.float64
    assert np.array_equal(d[1], flows[1].astype(float))
    for c in range(2):
        assert np.allclose(d[0, :, :, c], FO.regionfill(flows[0, :, :, c], masks[0, :, :, 0]))
