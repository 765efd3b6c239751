"""Poisson blending: oracle pins on CPU, HIP parity on the GPU.

CPU: `oracle.blend_oracle` reproduces the REFERENCE'S OWN tool/utils/Poisson_blend_img.py (which runs here: numpy + scipy only):
the equation system and UnfilledMask exactly, the blend within the reference's LSQR tolerance — committed golden
(tests/golden/blend_32x40.npz) and live when the reference is mounted.
GPU (-m gpu): `fgt_poisson_blend` against the oracle's least-squares solution (LSQR iterated to 1e-13).  UnfilledMask is index work:
equal.  The blend is a float solve: the bar is 1e-4 absolute on the 0..1 image scale at the pixels UnfilledMask does not flag
(the tool repaints the flagged ones; on singular components the two solvers need not agree); the reference itself stops LSQR at
1e-6 relative and lands within ~1e-5 of that solution.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import blend_oracle as BO
from oracle import reference_prop as RP
from util import GOLDEN

sys.path.insert(0, GOLDEN)
from make_golden_blend import blend_inputs  # noqa: E402


def _golden():
    z = np.load(os.path.join(GOLDEN, "blend_32x40.npz"))
    return {k: z[k] for k in z.files}


def test_oracle_matches_reference_golden():
    g = _golden()
    blend, unf = BO.poisson_blend(g["trg"], g["gx"], g["gy"], g["hole"], g["gmask"])
    assert np.array_equal(unf, g["unfilled"]) and 0 < unf.sum() < g["hole"].sum()
    assert np.abs(blend - g["blend"])[~unf].max() < 2e-5           # reference = LSQR at default tolerances
    loose, _ = BO.poisson_blend(g["trg"], g["gx"], g["gy"], g["hole"], g["gmask"], tight=False)
    assert np.abs(loose - g["blend"]).max() < 5e-6                   # same solver settings as the reference: same iterates


@pytest.mark.skipif(not RP.available(), reason="reference tree not mounted (GPU box)")
@pytest.mark.parametrize("case", [(24, 28, 1), (40, 36, 2), (17, 50, 3)])
def test_oracle_equals_reference_code_live(case):
    import importlib
    H, W, seed = case
    trg, gx, gy, hole, gmask = blend_inputs(H, W, seed)
    fn = RP.poisson_blend_fn()
    want_b, want_u = fn(trg.copy(), gx, gy, hole.copy(), gmask.copy())
    got_b, got_u = BO.poisson_blend(trg, gx, gy, hole, gmask, tight=False)
    assert np.array_equal(got_u, want_u.astype(bool))
    assert np.abs(got_b - want_b).max() < 5e-6
    mod = sys.modules.get(fn.__module__) or RP._import_from_tool("utils.Poisson_blend_img")
    A, b, _ = mod.solvePoisson(hole.copy(), gx, gy, trg, gmask.astype(np.float32), np.zeros(hole.shape, np.float32))
    A2, b2 = BO.equations(trg, gx, gy, hole, gmask)
    assert (A != A2).nnz == 0 and np.array_equal(b.astype(np.float64), b2)


# ------------------------------------------------------------------------------------------------ GPU
def _full(gx, gy, H, W):
    fx, fy = np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.float32)
    fx[:, : W - 1], fy[: H - 1] = gx, gy
    return fx, fy


@pytest.mark.gpu
def test_hip_blend_matches_oracle_on_reference_golden(dev):
    from fgt_amd.blending import Poisson_blend_img
    g = _golden()
    blend, unf = Poisson_blend_img(g["trg"], g["gx"], g["gy"], g["hole"], g["gmask"])          # reference signature
    want, wunf = BO.poisson_blend(g["trg"], g["gx"], g["gy"], g["hole"], g["gmask"])
    assert np.array_equal(unf, wunf) and np.array_equal(unf, g["unfilled"])
    d = np.abs(blend - want)[~wunf]
    print(f"[parity] poisson blend 32x40 golden: max |hip - lsq| {d.max():.2e}, |hip - reference lsqr| {np.abs(blend - g['blend'])[~wunf].max():.2e}")
    assert d.max() < 1e-4 and np.array_equal(blend[~g["hole"]], g["trg"][~g["hole"]])


@pytest.mark.gpu
def test_hip_blend_clip_432x240_matches_oracle(dev):
    """Three 240x432 frames with ~17 k-pixel holes in one call (different masks per frame), vs the oracle frame by frame."""
    from fgt_amd import ops
    H, W = 240, 432
    cases = [blend_inputs(H, W, s) for s in (5, 6, 7)]
    st = lambda k: torch.from_numpy(np.stack([np.ascontiguousarray(c[k]) for c in cases])).to(dev)
    fx = torch.from_numpy(np.stack([_full(c[1], c[2], H, W)[0] for c in cases])).to(dev)
    fy = torch.from_numpy(np.stack([_full(c[1], c[2], H, W)[1] for c in cases])).to(dev)
    blend, unf = ops.poisson_blend(st(0), fx, fy, st(3), st(4))
    blend, unf = blend.cpu().numpy(), unf.cpu().numpy()
    for i, (trg, gx, gy, hole, gmask) in enumerate(cases):
        want, wunf = BO.poisson_blend(trg, gx, gy, hole, gmask)
        assert np.array_equal(unf[i], wunf)
        d = np.abs(blend[i] - want)[~wunf]
        print(f"[parity] poisson blend 432x240 frame {i}: hole px {hole.sum()}, unfilled {wunf.sum()}, max |hip - lsq| {d.max():.2e}")
        assert d.max() < 1e-4


@pytest.mark.gpu
def test_hip_blend_edge_cases(dev):
    """Empty holes (nothing to solve), a hole covering a whole frame (no boundary equation anywhere: everything unfilled), and a frame wider
    than one scan chunk (W > 1024: the UnfilledMask row scan carries across chunks)."""
    from fgt_amd import ops
    rng = np.random.default_rng(3)
    H, W = 6, 1100
    trg = rng.random((3, H, W, 3)).astype(np.float32)
    gx = (rng.normal(size=(3, H, W, 3)) * 0.05).astype(np.float32)
    gy = (rng.normal(size=(3, H, W, 3)) * 0.05).astype(np.float32)
    hole = np.zeros((3, H, W), bool)
    gm = np.zeros((3, H, W), bool)
    hole[1] = True                                     # whole frame
    hole[2, 1:5, 3:1097] = True                        # one long hole spanning both scan chunks ...
    gm[2, 1:5, 600] = True                             # ... cut by a column of unknown gradients
    gm[2, 2, 100:1000] = True
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    blend, unf = ops.poisson_blend(t(trg), t(gx), t(gy), t(hole), t(gm))
    blend, unf = blend.cpu().numpy(), unf.cpu().numpy()
    assert np.array_equal(blend[0], trg[0]) and not unf[0].any()
    assert unf[1].all() and np.isfinite(blend[1]).all()
    want, wunf = BO.poisson_blend(trg[2], gx[2][:, : W - 1], gy[2][: H - 1], hole[2], gm[2])
    assert np.array_equal(unf[2], wunf) and 0 < wunf.sum() < hole[2].sum()
    assert np.abs(blend[2] - want)[~wunf].max() < 1e-4


@pytest.mark.gpu
def test_hip_blend_onchip_matches_multilaunch_and_oracle(dev):
    """csrc/solve_onchip.hip (one workgroup per (frame, channel), all CG iterations in one launch) vs the multi-launch kernels vs the exact
    least-squares solution: two 240x432 frames whose holes fit one workgroup (an ellipse of ~17 k px with a band of unknown gradients; a
    hole touching the image corner at an unaligned column), bit-reproducible, status = iterations used."""
    from fgt_amd import ops
    H, W = 240, 432
    yy, xx = np.mgrid[:H, :W]
    holes = [((yy - 120) / 70.0) ** 2 + ((xx - 201) / 80.0) ** 2 <= 1.0, np.zeros((H, W), bool)]
    holes[1][:80, :119] = True
    holes[1][50:120, 90:161] = True                     # bounding box 120 x 161 from the image corner: 4 920 strips, fits one workgroup
    cases = []
    for s, hole in enumerate(holes):
        trg, gx, gy, _, _ = blend_inputs(H, W, 20 + s)
        src = trg.copy()                                # blend_inputs zeroed its own hole: rebuild gradients for ours from a smooth image
        rng = np.random.default_rng(30 + s)
        from scipy.ndimage import gaussian_filter
        img = gaussian_filter(rng.normal(size=(H, W, 3)), (3, 3, 0))
        img = ((img - img.min()) / (img.max() - img.min())).astype(np.float32)
        gmask = np.zeros((H, W), bool)
        gmask[100:112, 150:230] = True
        gmask &= hole
        gx = np.diff(img, axis=1).astype(np.float32)
        gy = np.diff(img, axis=0).astype(np.float32)
        gx[gmask[:, :-1]] = 0
        gy[gmask[:-1, :]] = 0
        trg = img.copy()
        trg[hole] = 0
        cases.append((trg, gx, gy, hole, gmask))
    st = lambda k: torch.from_numpy(np.stack([np.ascontiguousarray(c[k]) for c in cases])).to(dev)
    fx = torch.from_numpy(np.stack([_full(c[1], c[2], H, W)[0] for c in cases])).to(dev)
    fy = torch.from_numpy(np.stack([_full(c[1], c[2], H, W)[1] for c in cases])).to(dev)
    a, ua = ops.poisson_blend(st(0), fx, fy, st(3), st(4), iters=4000, tol=1e-7, solver="onchip")
    info = ops.last_solver["poisson_blend"]
    assert info["solver"] == "onchip"
    status = info["status"].cpu().numpy()
    assert not (status & 1).any()
    a2, _ = ops.poisson_blend(st(0), fx, fy, st(3), st(4), iters=4000, tol=1e-7, solver="onchip")
    assert torch.equal(a, a2)
    b, ub = ops.poisson_blend(st(0), fx, fy, st(3), st(4), iters=4000, tol=1e-7, solver="multilaunch")
    assert torch.equal(ua, ub)
    a, b, ua = a.cpu().numpy(), b.cpu().numpy(), ua.cpu().numpy()
    for i, (trg, gx, gy, hole, gmask) in enumerate(cases):
        want, wunf = BO.poisson_blend(trg, gx, gy, hole, gmask)
        assert np.array_equal(ua[i], wunf)
        da, db = np.abs(a[i] - want)[~wunf].max(), np.abs(b[i] - want)[~wunf].max()
        print(f"[parity] poisson blend on-chip 432x240 frame {i}: hole px {hole.sum()}, iterations {[int(x) >> 1 for x in status[3 * i:3 * i + 3]]}, "
              f"max |on-chip - lsq| {da:.2e}, |multi-launch - lsq| {db:.2e}")
        assert da < 1e-4 and db < 1e-4
        assert np.array_equal(a[i][~hole], trg[~hole])
