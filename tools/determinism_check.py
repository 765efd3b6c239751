#!/usr/bin/env python
"""Run-to-run determinism of the FGT stage: N passes of ClipRunner over the bench clip, every composite compared bit for bit with the first
(eager and hipGraph replay).  Two of these side by side on one GPU (`... & ... & wait`) change every kernel's timing: a race shows up as a
differing composite.    python tools/determinism_check.py [--passes 6] [--graphs]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops  # noqa: E402
from fgt_amd.fgt_model import DEFAULT_CONFIG, Model  # noqa: E402
from fgt_amd.scheduler import ClipRunner  # noqa: E402
from fgt_amd.synth import synth_clip, synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=6)
ap.add_argument("--graphs", action="store_true")
ap.add_argument("--frames", type=int, default=80)
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
m = Model(dict(DEFAULT_CONFIG)).eval()
m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
m = m.to(dev)
fr, fl, ms = synth_clip(a.frames, 240, 432, seed=1234, device=dev)
r = ClipRunner(m, fr, fl, ms, use_graphs=a.graphs, window_batch=8, encode_chunk=40)
r.run()
ref = r.run().clone()
torch.cuda.synchronize()
bad = 0
for i in range(a.passes):
    c = r.run()
    torch.cuda.synchronize()
    d = (c != ref)
    n = int(d.sum())
    if n:
        bad += 1
        fr_ids = torch.nonzero(d.flatten(1).any(1)).flatten().tolist()
        print(f"pass {i}: {n} differing values, max |diff| {float((c - ref).abs().max())}, frames {fr_ids[:12]}")
print(f"pid {os.getpid()} graphs={a.graphs}: {a.passes} passes, {bad} differ from the first; checksum {float(ref.double().mean()):.6f}")
