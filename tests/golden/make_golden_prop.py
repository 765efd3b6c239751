"""Golden vectors of the flow-guided gradient propagation, produced by RUNNING THE REFERENCE'S OWN
tool/get_flowNN_gradient.py (authoring container only):

    python tests/golden/make_golden_prop.py

oracle/reference_prop.py imports the reference file with a stub `cv2` whose `remap` is the documented stand-in
`oracle.prop_oracle.remap_bilinear` (cv2 itself cannot be installed here); everything else that runs is the reference's code.
  prop_6x40x56.npz   6 frames, a box hole drifting 2 px / frame over flows of ~ +-8 px (chains through several frames, pixels
                     pushed out of the frame, round trips above and below the threshold of 5): inputs + outputs (gradients at the hole pixels, mask_tofill) for tab = 32
                     (OpenCV-style coordinate table) and tab = 0 (float bilinear)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_prop as RP  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def prop_inputs(N, H, W, seed, flow_scale=20.0, noise=2.0):
    """Seeded inputs shared with tests/test_prop_*.py: smooth flows, nearly inverse backward flows, a drifting box hole."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    sm = lambda a: gaussian_filter(a, (0, 3, 3, 0)).astype(np.float32)
    ff = sm(rng.normal(size=(N - 1, H, W, 2)) * flow_scale)
    fb = (-ff + sm(rng.normal(size=(N - 1, H, W, 2)) * noise)).astype(np.float32)
    mask = np.zeros((N, H, W), bool)
    bh, bw = int(H * 0.4), int(W * 0.4)
    for t in range(N):
        y0, x0 = H // 4 + t % 5, (W // 6 + 2 * t) % max(1, W - bw)
        mask[t, y0:y0 + bh, x0:x0 + bw] = True
    img = gaussian_filter(rng.normal(size=(N, H, W, 3)), (0, 2, 2, 0)).astype(np.float32)
    gx = np.concatenate((np.diff(img, axis=2), np.zeros((N, H, 1, 3), np.float32)), 2)
    gy = np.concatenate((np.diff(img, axis=1), np.zeros((N, 1, W, 3), np.float32)), 1)
    gx[mask] = 0
    gy[mask] = 0
    return gx, gy, mask, ff, fb


def main():
    gx, gy, mask, ff, fb = prop_inputs(6, 40, 56, seed=0)
    out = {"gx": gx, "gy": gy, "mask": mask, "flow_f": ff, "flow_b": fb}
    for tab in (32, 0):
        ox, oy, fill = RP.run_get_flownn_gradient(gx, gy, mask, ff, fb, thres=5.0, alpha=0.1, tab=tab)
        assert np.array_equal(ox[~mask], gx[~mask]) and np.array_equal(oy[~mask], gy[~mask])       # only hole pixels change: store those
        out[f"out_gx_tab{tab}"], out[f"out_gy_tab{tab}"], out[f"tofill_tab{tab}"] = ox[mask], oy[mask], fill
        print(f"tab={tab}: hole px {mask.sum()}, filled {(mask & ~fill).sum()}, unfilled {fill.sum()}")
    np.savez_compressed(os.path.join(OUT, "prop_6x40x56.npz"), **out)


if __name__ == "__main__":
    main()
