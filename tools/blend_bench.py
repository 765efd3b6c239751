#!/usr/bin/env python
"""Time fgt_poisson_blend on 80 frames of 432x240 with ~17 k-pixel holes (one call) and report the CG residual behaviour."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fgt_amd import ops  # noqa: E402
from make_golden_blend import blend_inputs  # noqa: E402

dev = torch.device("cuda:0")
H, W, N = 240, 432, 80
c = [blend_inputs(H, W, s) for s in range(4)]


def full(g, axis):
    f = np.zeros((H, W, 3), np.float32)
    if axis == 1:
        f[:, : W - 1] = g
    else:
        f[: H - 1] = g
    return f


st = lambda arrs: torch.from_numpy(np.stack([arrs[i % 4] for i in range(N)])).to(dev)
trg, hole, gm = st([x[0] for x in c]), st([x[3] for x in c]), st([x[4] for x in c])
gx, gy = st([full(x[1], 1) for x in c]), st([full(x[2], 0) for x in c])
for iters in (500, 1500, 3000):
    ops.poisson_blend(trg, gx, gy, hole, gm, iters=iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, unf = ops.poisson_blend(trg, gx, gy, hole, gm, iters=iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref, _ = ops.poisson_blend(trg[:4], gx[:4], gy[:4], hole[:4], gm[:4], iters=12000, tol=0.0)
    d = (out[:4] - ref).abs()[~unf[:4]].max().item()
    print(f"poisson blend {N} frames 432x240 ({int(hole[0].sum())} hole px / frame), {iters} CG iterations: {dt * 1e3:.1f} ms per clip, max |x - x(12000 its)| on filled pixels {d:.2e}")
