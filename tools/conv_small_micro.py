#!/usr/bin/env python
"""The Cout <= 4 direct convolution (csrc/conv_direct.hip: conv3x3_tiled_kernel) at its two shapes on the path — FGT's decoder.final 64 -> 3 at
240x432 (40 frames) and RAFT's flow head 256 -> 2 at 60x108 (32 pairs) — event-timed, with the algorithmic bytes (input once + output); target of
`rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE`.   FGT_CONV_SMALL_C32=0: the 16-channel chunks of rounds 1-5."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402
from fgt_amd.ops import PackedConv  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.DEFAULT_CONV_PRECISION = "bf16x3"
g = torch.Generator().manual_seed(0)
for name, (N, H, W, Cin, Cout) in {"fgt decoder.final": (40, 240, 432, 64, 3), "raft flow head": (32, 60, 108, 256, 2)}.items():
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    pc = PackedConv((torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev), torch.zeros(Cout).to(dev))
    fn = lambda: ops.conv2d(x, pc, pad=1)
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = N * H * W * (Cin + Cout) * 4
    print(f"{name}: {ms * 1e3:.1f} us per launch, {by / 1e6:.1f} MB algorithmic, {by / ms / 1e6:.0f} GB/s (C32={os.environ.get('FGT_CONV_SMALL_C32', '1')})")
