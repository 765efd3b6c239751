"""fgt_laplace_fill (batched CG diffusion fill) on a real MI355X vs the reference goldens / the scipy oracle.
The reference solves each map exactly in float64; the GPU path iterates in fp32 to a relative residual of 1e-6, and its consumer
(LAFC) takes float32 inputs: tolerance 1e-4 of the map's value range (flows are in pixels), unmasked pixels bit-identical."""
import os

import numpy as np
import pytest
import torch

from oracle import fill_oracle as FO
from util import GOLDEN

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize("name", ["blobs", "borders", "empty", "pixels", "large"])
def test_fill_matches_reference_golden(name, dev):
    from fgt_amd import ops
    g = np.load(os.path.join(GOLDEN, f"fill_{name}.npz"))
    I = torch.from_numpy(g["I"]).to(dev)[None]
    m = torch.from_numpy(g["mask"]).to(dev)[None]
    out = ops.laplace_fill(I, m, iters=600, tol=1e-7)[0].cpu().numpy()
    hole = g["mask"] != 0
    assert np.array_equal(out[~hole], g["I"][~hole])
    scale = float(np.abs(g["out"]).max())
    err = float(np.abs(out - g["out"]).max())
    print(f"[parity] laplace_fill {name}: max_abs={err:.3e} of range {scale:.3e}")
    assert err <= 1e-4 * scale
    assert FO.residual(out.astype(np.float64), g["I"], g["mask"]) <= 2e-4 * scale


def test_fill_clip_sized_batch_vs_oracle(dev):
    """240x432 maps with a large object-like hole (~17 k px) and border-touching holes; 6 maps sharing 3 masks (b % n_masks),
    fixed default iteration count; bit-reproducible."""
    from fgt_amd import ops
    rng = np.random.default_rng(3)
    H, W, n = 240, 432, 3
    yy, xx = np.mgrid[:H, :W]
    masks = np.zeros((n, H, W), dtype=np.uint8)
    masks[0] = ((yy - 120) / 70.0) ** 2 + ((xx - 200) / 80.0) ** 2 <= 1.0          # ~17.6 k px blob
    masks[1, 60:200, :50] = 1
    masks[1, :30, 300:] = 1
    masks[2] = masks[0] | masks[1]
    maps = (np.cumsum(rng.standard_normal((2 * n, H, W)), axis=2) * 0.5).astype(np.float32)
    a = ops.laplace_fill(torch.from_numpy(maps).to(dev), torch.from_numpy(masks).to(dev))
    b = ops.laplace_fill(torch.from_numpy(maps).to(dev), torch.from_numpy(masks).to(dev))
    assert torch.equal(a, b)
    a = a.cpu().numpy()
    for i in range(2 * n):
        ref = FO.regionfill(maps[i], masks[i % n])
        scale = float(np.abs(ref).max())
        err = float(np.abs(a[i] - ref).max())
        print(f"[parity] laplace_fill 240x432 map {i} (mask {i % n}, {int(masks[i % n].sum())} px): max_abs={err:.3e} of range {scale:.3e}")
        assert err <= 1e-4 * scale
        assert np.array_equal(a[i][masks[i % n] == 0], maps[i][masks[i % n] == 0])


def test_diffusion_pipeline_layout(dev):
    """flow_pipeline.diffusion: [1,2,t,H,W] flows + [1,1,t,H,W] masks, the layout complete_flows consumes (tool/video_inpainting.py:42-51,355)."""
    from fgt_amd import flow_pipeline as FP
    rng = np.random.default_rng(5)
    t, H, W = 4, 48, 64
    flows = rng.standard_normal((t, H, W, 2)).astype(np.float32) * 2
    masks = np.zeros((t, H, W, 1), dtype=np.uint8)
    for i in range(t):
        masks[i, 10 + i:30 + i, 20:45] = 1
    ref = FO.diffusion(flows, masks)                                             # [t,H,W,2]
    fl = torch.from_numpy(flows).permute(3, 0, 1, 2)[None].contiguous().to(dev)   # np2tensor layout [1,2,t,H,W]
    mk = torch.from_numpy(masks).permute(3, 0, 1, 2)[None].float().contiguous().to(dev)
    out = FP.diffusion(fl, mk)
    assert out.shape == fl.shape
    got = out[0].permute(1, 2, 3, 0).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


# ---- one workgroup per problem, all iterations inside one launch (csrc/solve_onchip.hip) vs the multi-launch kernels
def _ellipse(H, W, cy, cx, ry, rx):
    yy, xx = np.mgrid[:H, :W]
    return (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).astype(np.uint8)


def test_mask_bbox(dev):
    from fgt_amd import ops
    m = np.zeros((4, 40, 52), np.uint8)
    m[0, 3:17, 9:33] = 1
    m[1, 39, 51] = 1
    m[2, 0, 0] = 1
    m[2, 20, 30] = 1                                   # mask 3 stays empty
    bb = ops.mask_bbox(torch.from_numpy(m).to(dev)).cpu().tolist()
    assert bb[0] == [3, 9, 16, 32] and bb[1] == [39, 51, 39, 51] and bb[2] == [0, 0, 20, 30]
    assert bb[3][2] < bb[3][0]


@pytest.mark.parametrize("case", ["ellipse_17k", "border_boxes", "unaligned_thin", "single_pixels", "two_masks_one_empty"])
def test_fill_onchip_matches_multilaunch_and_oracle(case, dev):
    """The on-chip solver runs the same CG iteration as the multi-launch kernels (other reduction order): both within 1e-4 of the
    direct solve, unmasked pixels untouched, bit-reproducible, status = iterations used (and never 'did not fit')."""
    from fgt_amd import ops
    rng = np.random.default_rng(11)
    H, W = 240, 432
    masks = np.zeros((2, H, W), np.uint8)
    if case == "ellipse_17k":
        masks[0] = _ellipse(H, W, 120, 200, 70, 80)
        masks[1] = _ellipse(H, W, 110, 260, 60, 75)
    elif case == "border_boxes":                       # holes touching every image border (n(p) = 3 / 2), box corners at (0, 0) and (H-1, W-1)
        masks[0, :40, :60] = 1
        masks[0, 30:70, 50:120] = 1
        masks[1, 200:, 380:] = 1
        masks[1, 180:210, 300:400] = 1
    elif case == "unaligned_thin":                     # box starts off a multiple of 4 columns, 1-pixel-wide parts
        masks[0, 50:130, 101:103] = 1
        masks[0, 90, 90:190] = 1
        masks[1, 10:200, 333] = 1
    elif case == "single_pixels":
        masks[0, 5, 7] = 1
        masks[0, 9, 7] = 1
        masks[1, 100, 431] = 1
    else:
        masks[0] = _ellipse(H, W, 100, 100, 40, 50)    # mask 1 empty: those maps are copied
    maps = (np.cumsum(rng.standard_normal((4, H, W)), axis=2) * 0.5).astype(np.float32)
    tm, tk = torch.from_numpy(maps).to(dev), torch.from_numpy(masks).to(dev)
    a = ops.laplace_fill(tm, tk, iters=3000, tol=1e-7, solver="onchip")
    st = ops.last_solver["laplace_fill"]
    assert st["solver"] == "onchip"
    status = st["status"].cpu().numpy()
    assert not (status & 1).any(), "a problem did not fit its workgroup"
    assert torch.equal(a, ops.laplace_fill(tm, tk, iters=3000, tol=1e-7, solver="onchip"))
    b = ops.laplace_fill(tm, tk, iters=3000, tol=1e-7, solver="multilaunch")
    assert ops.last_solver["laplace_fill"]["solver"] == "multilaunch"
    a, b = a.cpu().numpy(), b.cpu().numpy()
    for i in range(4):
        mk = masks[i % 2]
        assert np.array_equal(a[i][mk == 0], maps[i][mk == 0])
        ref = FO.regionfill(maps[i], mk) if mk.any() else maps[i]
        scale = float(np.abs(ref).max())
        ea, eb = float(np.abs(a[i] - ref).max()), float(np.abs(b[i] - ref).max())
        print(f"[parity] laplace_fill on-chip {case} map {i} ({int(mk.sum())} px, {int(status[i]) >> 1} iterations): "
              f"max_abs {ea:.3e} (multi-launch {eb:.3e}) of range {scale:.3e}")
        assert ea <= 1e-4 * scale and eb <= 1e-4 * scale


def test_fill_onchip_falls_back_when_the_box_is_too_large(dev):
    """A hole whose bounding box exceeds one workgroup's capacity: 'auto' takes the multi-launch kernels, 'onchip' refuses loudly."""
    from fgt_amd import ops
    H, W = 240, 432
    masks = np.zeros((1, H, W), np.uint8)
    masks[0, 5:230, 10:420] = np.random.default_rng(0).random((225, 410)) > 0.5
    maps = np.random.default_rng(1).standard_normal((2, H, W)).astype(np.float32)
    tm, tk = torch.from_numpy(maps).to(dev), torch.from_numpy(masks).to(dev)
    out = ops.laplace_fill(tm, tk, iters=200, tol=1e-6)
    assert ops.last_solver["laplace_fill"]["solver"] == "multilaunch" and bool(torch.isfinite(out).all())
    with pytest.raises(RuntimeError, match="does not fit"):
        ops.laplace_fill(tm, tk, iters=10, solver="onchip")
