"""CPU restatement of the reference's flow "diffusion" fill (TEST INFRASTRUCTURE ONLY — the product never imports oracle/).

Follows tool/utils/region_fill.py:7-117 (`regionfill` at factor = 1, the only way the tool calls it) and `diffusion`,
tool/video_inpainting.py:42-51.  numpy float64 + scipy's sparse direct solver, like the reference.
Pinned: tests/golden/fill_*.npz are outputs of the reference's own `regionfill`, imported from /root/reference with a
stub `cv2` (tests/golden/make_golden_fill.py) — at factor 1 the reference only needs cv2 for an identity resize and a 3x3
cross dilation.
"""
import numpy as np
from scipy import sparse
from scipy.sparse.linalg import spsolve


def num_neighbors(H, W):
    """region_fill.py:104-117: in-image 4-neighbour count (4 interior, 3 border, 2 corner)."""
    n = np.full((H, W), 4.0)
    n[0, :] -= 1
    n[-1, :] -= 1
    n[:, 0] -= 1
    n[:, -1] -= 1
    return n


def right_side(I, mask):
    """region_fill.py:19-24,66-101: sum over the in-image 4-neighbours that lie on the mask perimeter (= are not masked;
    a 4-neighbour of a masked pixel is either masked or in the cross-dilated ring)."""
    H, W = I.shape
    per = np.where(mask, 0.0, I)
    rs = np.zeros((H, W))
    rs[1:, :] += per[:-1, :]
    rs[:-1, :] += per[1:, :]
    rs[:, 1:] += per[:, :-1]
    rs[:, :-1] += per[:, 1:]
    return rs


def regionfill(I, mask):
    """region_fill.py:7-63 with factor = 1.0: Laplace equation on the masked pixels, Dirichlet data from the unmasked ones."""
    I = np.asarray(I)
    mask = np.asarray(mask) != 0
    if not mask.any():
        return I.copy()
    H, W = I.shape
    If = I.astype(float)
    idx = -np.ones((H, W), dtype=np.int64)
    ys, xs = np.where(mask)
    idx[ys, xs] = np.arange(ys.size)
    rows, cols, vals = [np.arange(ys.size)], [np.arange(ys.size)], [num_neighbors(H, W)[ys, xs]]
    for dy, dx in ((-1, 0), (0, 1), (1, 0), (0, -1)):
        ny, nx = ys + dy, xs + dx
        ok = (ny >= 0) & (ny < H) & (nx >= 0) & (nx < W)
        nb = np.full(ys.size, -1, dtype=np.int64)
        nb[ok] = idx[ny[ok], nx[ok]]
        sel = nb >= 0
        rows.append(np.arange(ys.size)[sel])
        cols.append(nb[sel])
        vals.append(-np.ones(sel.sum()))
    D = sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols)))).tocsr()
    sol = spsolve(D, right_side(If, mask)[ys, xs])
    out = If.copy()
    out[ys, xs] = sol
    out[~mask] = I[~mask]                       # region_fill.py:15
    return out


def diffusion(flows, masks):
    """tool/video_inpainting.py:42-51: flows [t, H, W, 2], masks [t, H, W, 1] -> [t, H, W, 2] (float64, like the reference)."""
    out = np.zeros(flows.shape)
    for i in range(flows.shape[0]):
        out[i, :, :, 0] = regionfill(flows[i, :, :, 0], masks[i, :, :, 0])
        out[i, :, :, 1] = regionfill(flows[i, :, :, 1], masks[i, :, :, 0])
    return out


def residual(x, I, mask):
    """max |n x - sum masked-neighbour x - rhs| over masked pixels (a size-independent property test of any solver)."""
    mask = np.asarray(mask) != 0
    H, W = x.shape
    xm = np.where(mask, x, 0.0)
    lap = num_neighbors(H, W) * x
    lap[1:, :] -= xm[:-1, :]
    lap[:-1, :] -= xm[1:, :]
    lap[:, 1:] -= xm[:, :-1]
    lap[:, :-1] -= xm[:, 1:]
    r = lap - right_side(np.asarray(I, dtype=float), mask)
    return float(np.abs(r[mask]).max()) if mask.any() else 0.0
