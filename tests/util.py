"""Shared test helpers: seeded inputs (same recipes as tests/golden/make_golden.py) and error metrics."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fgt_inputs(H, W, t, seed):
    g = torch.Generator().manual_seed(seed)
    fr = torch.rand(1, t, 3, H, W, generator=g) * 2 - 1
    ms = (torch.rand(1, t, 1, H // 8, W // 8, generator=g) > 0.7).float()
    ms = torch.nn.functional.interpolate(ms.view(t, 1, H // 8, W // 8), size=(H, W), mode="nearest").view(1, t, 1, H, W)
    fl = torch.randn(1, t, 2, H, W, generator=g)
    return fr * (1 - ms), fl, ms


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


def max_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).abs().max().item()


def rel_err(a, b):
    """max |a-b| relative to the reference's max magnitude."""
    b = b.detach().float().cpu()
    return max_err(a, b) / max(b.abs().max().item(), 1e-30)


def report(name, a, b):
    e, r = max_err(a, b), rel_err(a, b)
    print(f"[parity] {name}: max_abs={e:.3e} rel_to_max={r:.3e} ref_max={b.abs().max().item():.3e}")
    return e, r
