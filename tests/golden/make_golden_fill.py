"""Golden vectors for the diffusion fill: run the REFERENCE's own regionfill (tool/utils/region_fill.py, imported from
/root/reference) on seeded inputs.  cv2 is not installed here; at factor = 1.0 (the only value the tool uses) the reference needs
it for three things that are restated exactly in the stub below: resize with fx = fy = 1 or to the same size (identity),
getStructuringElement(MORPH_CROSS, (3, 3)) and dilate with that cross.   python tests/golden/make_golden_fill.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _cv2_stub():
    cv2 = types.ModuleType("cv2")
    cv2.MORPH_CROSS = 1

    def resize(a, dsize, fx=None, fy=None):
        if tuple(dsize) == (0, 0):
            assert fx == 1.0 and fy == 1.0, "stub only restates the identity resize"
        else:
            assert tuple(dsize) == (a.shape[1], a.shape[0]), "stub only restates the identity resize"
        return np.array(a, dtype=float, copy=True)

    def get_se(shape, ksize):
        assert shape == cv2.MORPH_CROSS and tuple(ksize) == (3, 3)
        return np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.uint8)

    def dilate(a, k):
        out = a.copy()
        out[1:, :] = np.maximum(out[1:, :], a[:-1, :])
        out[:-1, :] = np.maximum(out[:-1, :], a[1:, :])
        out[:, 1:] = np.maximum(out[:, 1:], a[:, :-1])
        out[:, :-1] = np.maximum(out[:, :-1], a[:, 1:])
        return out

    cv2.resize, cv2.getStructuringElement, cv2.dilate = resize, get_se, dilate
    return cv2


def blobs(rng, H, W, n, rmax):
    m = np.zeros((H, W), dtype=bool)
    yy, xx = np.mgrid[:H, :W]
    for _ in range(n):
        cy, cx, ry, rx = rng.integers(0, H), rng.integers(0, W), rng.integers(2, rmax), rng.integers(2, rmax)
        m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    return m


def main():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("cv2", _cv2_stub())
    sys.path.insert(0, os.path.join(REF, "tool"))
    from utils import region_fill as rf            # the reference implementation
    rng = np.random.default_rng(7)
    cases = {}
    H, W = 40, 56
    I = (rng.standard_normal((H, W)) * 3).astype(np.float32)
    cases["blobs"] = (I, blobs(rng, H, W, 3, 12))
    m = np.zeros((H, W), dtype=bool); m[:9, :11] = True; m[30:, 50:] = True; m[15:22, :] = True      # corner, corner, full-width band
    cases["borders"] = (I, m)
    cases["empty"] = (I, np.zeros((H, W), dtype=bool))
    m = np.zeros((H, W), dtype=bool); m[5, 7] = True; m[20:22, 30] = True                              # single pixels
    cases["pixels"] = (I, m)
    H, W = 96, 128
    I2 = np.cumsum(rng.standard_normal((H, W)), axis=1).astype(np.float32)
    cases["large"] = (I2, blobs(rng, H, W, 4, 30))
    for name, (img, mask) in cases.items():
        out = rf.regionfill(img.copy(), mask.astype(np.uint8))
        np.savez_compressed(os.path.join(HERE, f"fill_{name}.npz"), I=img, mask=mask.astype(np.uint8), out=np.asarray(out, dtype=np.float64))
        print(name, img.shape, int(mask.sum()), "masked px, out range", float(np.min(out)), float(np.max(out)))


if __name__ == "__main__":
    main()
