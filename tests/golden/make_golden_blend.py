"""Golden vectors of the Poisson blending step, produced by RUNNING THE REFERENCE'S OWN tool/utils/Poisson_blend_img.py
(numpy + scipy only; the module-level `import cv2` is satisfied by the stub of oracle/reference_prop.py and never called):

    python tests/golden/make_golden_blend.py

  blend_32x40.npz   one 32x40 frame: a blob hole, a gradient mask that cuts part of it off (so UnfilledMask is non-empty), gradients of
                    a smooth image + noise; the reference's imgBlend (LSQR at its default tolerances) and UnfilledMask
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_prop as RP  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def blend_inputs(H, W, seed):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    img = gaussian_filter(rng.normal(size=(H, W, 3)), (3, 3, 0))
    img = ((img - img.min()) / (img.max() - img.min())).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    hole = ((yy - H * 0.5) ** 2 / (H * 0.3) ** 2 + (xx - W * 0.45) ** 2 / (W * 0.3) ** 2) < 1
    hole[: H // 6, : W // 5] = True                                   # a second component touching the image corner
    gmask = np.zeros((H, W), bool)
    gmask[int(H * 0.45):int(H * 0.6), int(W * 0.3):int(W * 0.55)] = True    # propagation left these gradients unknown
    gmask[int(H * 0.3), int(W * 0.2):int(W * 0.7)] = True                 # a cut line
    gmask &= hole
    src = np.clip(img + gaussian_filter(rng.normal(size=(H, W, 3)), (2, 2, 0)).astype(np.float32) * 0.5, 0, 1)
    gx = np.diff(src, axis=1).astype(np.float32)
    gy = np.diff(src, axis=0).astype(np.float32)
    gx[gmask[:, :-1]] = 0
    gy[gmask[:-1, :]] = 0
    trg = img.copy()
    trg[hole] = 0
    return trg, gx, gy, hole, gmask


def main():
    fn = RP.poisson_blend_fn()
    trg, gx, gy, hole, gmask = blend_inputs(32, 40, 0)
    blend, unfilled = fn(trg.copy(), gx, gy, hole.copy(), gmask.copy())
    print("hole px", hole.sum(), "unfilled", unfilled.sum(), "blend range", blend[hole].min(), blend[hole].max())
    np.savez_compressed(os.path.join(OUT, "blend_32x40.npz"), trg=trg, gx=gx, gy=gy, hole=hole, gmask=gmask, blend=blend.astype(np.float32),
                        unfilled=unfilled.astype(bool))


if __name__ == "__main__":
    main()
