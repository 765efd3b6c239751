"""Poisson blending behind the reference's call signature.

`Poisson_blend_img(imgTrg, imgSrc_gx, imgSrc_gy, holeMask, gradientMask)` is what tool/video_inpainting.py:651-657 calls per frame
(tool/utils/Poisson_blend_img.py:19): numpy arrays imgTrg [H,W,3], imgSrc_gx [H,W-1,3], imgSrc_gy [H-1,W,3], masks [H,W].
`poisson_blend_clip` is the device-tensor entry point that blends every frame of a clip in ONE `fgt_poisson_blend` call
(csrc/poisson_blend.hip: batched conjugate gradients on the normal equations of the reference's least-squares system).
The reference's LSQR stops at atol = btol = 1e-6; the CG here runs to a relative residual of 1e-7 (measured distance to the exact
least-squares solution: tests/test_blend_pinned.py).  `edge` (always zero in the tool) is not supported.
"""
import numpy as np
import torch

from . import ops


BLEND_ITERS = 2000           # iteration cap of the conjugate gradients (a problem stops at tol * |r0|)


def poisson_blend_clip(target, gradient_x, gradient_y, hole, gradient_mask, iters=BLEND_ITERS, tol=1e-7, rank=0, world=1, group=None, bounds=None):
    """target, gradient_x, gradient_y [N,H,W,3] fp32 device tensors (gradient_x[..., x, :] = I[x+1] - I[x]); hole, gradient_mask
    [N,H,W].  Returns (blend [N,H,W,3], UnfilledMask [N,H,W] bool).  world > 1: frames are independent problems
    (tool/video_inpainting.py:644-682 loops over them): block-sharded, one all-gather of the blended frames + unfilled masks.
    `bounds` = ops.hole_bounds(hole) of the clip (no host sync in this call)."""
    if world == 1:
        return ops.poisson_blend(target, gradient_x, gradient_y, hole, gradient_mask, iters, tol, bounds=bounds)
    from .flow_pipeline import gather_blocks, shard_range
    N, H, W, _ = target.shape
    lo, hi = shard_range(N, rank, world)
    if hi > lo:
        bl, unf = ops.poisson_blend(target[lo:hi], gradient_x[lo:hi], gradient_y[lo:hi], hole[lo:hi], gradient_mask[lo:hi], iters, tol, bounds=bounds)
        loc = torch.cat([bl, unf.float()[..., None]], -1)                       # [cnt, H, W, 4]: one collective for both outputs
    else:
        loc = torch.zeros(0, H, W, 4, dtype=torch.float32, device=target.device)
    full = gather_blocks(loc, N, rank, world, group)
    return full[..., :3].contiguous(), full[..., 3] != 0


def Poisson_blend_img(imgTrg, imgSrc_gx, imgSrc_gy, holeMask, gradientMask=None, edge=None, device="cuda", iters=BLEND_ITERS, tol=1e-7):
    """Reference signature / layouts (tool/utils/Poisson_blend_img.py:19-75) for one frame."""
    if isinstance(edge, np.ndarray) and edge.any():
        raise NotImplementedError("edge constraints are not built (the tool always passes edge=None)")
    H, W, _ = imgTrg.shape
    dev = torch.device(device)
    gm = np.zeros((H, W), bool) if not isinstance(gradientMask, np.ndarray) else gradientMask != 0
    gx = np.zeros((H, W, 3), np.float32)
    gy = np.zeros((H, W, 3), np.float32)
    gx[:, : W - 1] = imgSrc_gx
    gy[: H - 1] = imgSrc_gy
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)[None]
    blend, unf = poisson_blend_clip(t(imgTrg.astype(np.float32)), t(gx), t(gy), t(np.asarray(holeMask) != 0), t(gm), iters, tol)
    return blend[0].cpu().numpy(), unf[0].cpu().numpy()
