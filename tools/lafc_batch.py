#!/usr/bin/env python
"""LAFC completion of an 80-frame direction with 8 / 16 / 32 pivots per call (flow_pipeline.complete_flows): ms per flow, bit-equality.
    python tools/lafc_batch.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import lafc_model, ops, flow_pipeline
from fgt_amd.synth import synth_state_dict
ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
m.load_state_dict(synth_state_dict(m.state_dict(), seed=0, mode="kaiming"), strict=True)
m = m.to(dev)
g = torch.Generator().manual_seed(0)
N, H, W = 80, 240, 432
flows = torch.randn(1, 2, N - 1, H, W, generator=g).to(dev)
masks = (torch.rand(1, 1, N - 1, H // 8, W // 8, generator=g) > 0.7).float().repeat_interleave(8, 3).repeat_interleave(8, 4).to(dev)
diff = flows * (1 - masks)
ref = None
for b in (8, 16, 32, 8, 16):
    for _ in range(2):
        out = flow_pipeline.complete_flows(m, flows, masks, diffused=diff, batch=b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = flow_pipeline.complete_flows(m, flows, masks, diffused=diff, batch=b)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    o = out[0] if isinstance(out, (tuple, list)) else out
    if ref is None: ref = o.clone()
    print(f"LAFC {N-1} flows, {b} pivots per call: {dt*1e3/(N-1):.3f} ms per flow, bit-equal to batch 8: {torch.equal(o, ref)}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GB")
