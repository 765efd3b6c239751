"""CPU oracle for the tool's Poisson blending step (TEST INFRASTRUCTURE ONLY).

numpy / scipy restatement of `Poisson_blend_img` + `solvePoisson` + `constructEquation`
(tool/utils/Poisson_blend_img.py:19-244; call site tool/video_inpainting.py:644-682): the over-determined system with one equation
per (hole pixel, 4-neighbour) whose connecting gradient is valid, solved in the least-squares sense, and the raster-order
`UnfilledMask` reachability.  The reference solves with scipy's LSQR at its DEFAULT tolerances (atol = btol = 1e-6: an approximate
solution); `tight=True` here iterates LSQR to 1e-13 so that the oracle is the least-squares solution itself.  Pinned on outputs of
the reference's own function (tests/golden/blend_*.npz via tests/golden/make_golden_blend.py; the reference file runs here with
numpy + scipy only): `UnfilledMask` and the equation system exactly, the blend within the reference's own LSQR tolerance.
"""
import numpy as np
from scipy import sparse
from scipy.sparse.linalg import lsqr


def equations(trg, gx, gy, hole, gmask):
    """The reference's (A, b) (Poisson_blend_img.py:77-141, 171-244) with rows in the reference's order: for n in right, down, left, up:
    first the boundary equations of all hole pixels, then the non-boundary ones.  gx [H,W-1,C], gy [H-1,W,C] like the tool's slices."""
    H, W = hole.shape
    hole = hole.astype(bool)
    gm = np.asarray(gmask) != 0
    pi, pj = np.nonzero(hole)
    pind = pi * W + pj
    rows, cols, vals, rhs = [], [], [], []
    e = 0
    for n, (dy, dx) in enumerate(((0, 1), (1, 0), (0, -1), (-1, 0))):
        qi, qj = pi + dy, pj + dx
        inside = (qi >= 0) & (qi < H) & (qj >= 0) & (qj < W)
        qic, qjc = np.where(inside, qi, 0), np.where(inside, qj, 0)
        if n == 0 or n == 1:
            have = ~gm[pi, pj]
        elif n == 2:
            have = ~gm[pi, pj - 1]          # (pj = 0 wraps like the reference's numpy index; `inside` is False there)
        else:
            have = ~gm[pi - 1, pj]
        ok = inside & have
        if n == 0:
            r = -gx[pi, np.minimum(pj, W - 2)]
        elif n == 2:
            r = gx[pi, np.maximum(pj - 1, 0)]
        elif n == 1:
            r = -gy[np.minimum(pi, H - 2), pj]
        else:
            r = gy[np.maximum(pi - 1, 0), pj]
        qhole = hole[qic, qjc]
        bnd = ok & ~qhole
        k = bnd.sum()
        rows.append(np.arange(e, e + k)); cols.append(pind[bnd]); vals.append(np.ones(k)); rhs.append(r[bnd] + trg[qic[bnd], qjc[bnd]])
        e += k
        nb = ok & qhole
        k = nb.sum()
        rows += [np.arange(e, e + k), np.arange(e, e + k)]
        cols += [pind[nb], (qic * W + qjc)[nb]]
        vals += [np.ones(k), -np.ones(k)]
        rhs.append(r[nb])
        e += k
    A = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(e, H * W))
    return A, np.concatenate(rhs, 0).astype(np.float64)


def unfilled_mask(hole, gmask):
    """Poisson_blend_img.py:143-168: pixels the two raster sweeps cannot reach through valid gradients."""
    H, W = hole.shape
    tl = hole.astype(np.uint8).copy()
    br = hole.astype(np.uint8).copy()
    gm = np.asarray(gmask) != 0
    pi, pj = np.nonzero(hole)
    for i, j in zip(pi, pj):
        if (i >= 1 and tl[i - 1, j] == 0 and not gm[i - 1, j]) or (j >= 1 and tl[i, j - 1] == 0 and not gm[i, j - 1]):
            tl[i, j] = 0
    for i, j in zip(pi[::-1], pj[::-1]):
        if (i + 1 <= H - 1 and br[i + 1, j] == 0 and not gm[i, j]) or (j + 1 <= W - 1 and br[i, j + 1] == 0 and not gm[i, j]):
            br[i, j] = 0
    return (tl * br).astype(bool)


def poisson_blend(trg, gx, gy, hole, gmask, tight=True):
    """trg [H,W,C] float32, gx [H,W-1,C], gy [H-1,W,C], hole / gmask [H,W] -> (blend [H,W,C] float32, UnfilledMask [H,W] bool)."""
    H, W, C = trg.shape
    A, b = equations(trg, gx, gy, hole, gmask)
    rec = np.zeros((H, W, C), np.float32)
    kw = dict(atol=1e-13, btol=1e-13, iter_lim=20 * H * W) if tight else {}
    for c in range(C):
        rec[:, :, c] = lsqr(A, b[:, c], **kw)[0].reshape(H, W)
    hm = hole.astype(bool)[..., None]
    return np.where(hm, rec, trg).astype(np.float32), unfilled_mask(hole, gmask)
