#!/usr/bin/env python
"""Time fgt_flow_propagate on the bench clip's geometry (432x240x80, ~17 k-pixel holes) next to the CPU oracle."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fgt_amd import ops  # noqa: E402
from make_golden_prop import prop_inputs  # noqa: E402
from oracle import prop_oracle as PO  # noqa: E402

dev = torch.device("cuda:0")
gx, gy, mask, ff, fb = prop_inputs(80, 240, 432, 11, flow_scale=12.0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
args = (t(gx), t(gy), t(mask), t(ff), t(fb))
for _ in range(2):
    ops.flow_propagate(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = ops.flow_propagate(*args)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
PO.get_flownn_gradient(gx, gy, mask, ff, fb)
cpu = time.perf_counter() - t0
n = 80 * 240 * 432
print(f"flow_propagate 432x240x80, {int(mask.sum())} hole px: GPU {gpu * 1e3:.2f} ms per clip ({n * 2 * 3 * 4 * 4 / gpu / 1e9:.0f} GB/s of gradient in+out traffic equivalent), "
      f"CPU oracle (vectorised numpy) {cpu:.2f} s  [the reference's own loop needs minutes]")
