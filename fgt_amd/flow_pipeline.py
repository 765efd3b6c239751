"""Clip-level flow stage restructured for the MI355X — SURVEY §8f ranks 2 and 4: the RAFT driver
(tool/video_inpainting.py:233-288, `calculate_flow`), the diffusion fill (:42-51) and the LAFC loop (:341-386, `complete_flow`).

The reference calls RAFT once per adjacent pair and direction: 2(N-1) calls of batch 1, every frame is pushed through the
feature encoder up to 4 times and through the context encoder twice, and at 432x240 a pair is only 1620 query pixels
(13 M-tiles of work per conv: the GPU is mostly idle).  Here
  * fnet / cnet run ONCE per frame (InstanceNorm is per sample and BatchNorm is in eval mode, so this is exact),
  * pairs are processed `batch` at a time (forward and backward pairs in the same batch), which fills the machine.
Outputs are the reference's per-pair `flow_up` fields in its order.
"""
import torch

from . import ops


RAFT_PAIR_BATCH = 64         # pairs per RAFT refinement batch (forward and backward pairs mixed): the per-iteration convs have 1 620 (432x240) /
#                              6 480 (864x480) output pixels per pair — 8 pairs left 1.6 rounds of tiles on 256 CUs.  Measured (profiles/r03_run2_raft_batch.txt):
#                              8 -> 32 pairs: 1.88 -> 1.16 ms per pair at 432x240, 4.34 -> 3.93 at 864x480, bit-identical flows, 10.6 GB peak;
#                              32 -> 64 (end of round 3, tap-reusing convs): 1.02 -> 0.95 and 3.35 -> 3.27, bit-identical, 22 GB peak
LAFC_PIVOT_BATCH = 16        # pivots per LAFC call (tools/lafc_batch.py: 8 -> 0.718, 16 -> 0.69, 32 -> 0.688 ms per flow at 432x240, bit-equal; 2.9 GB peak at 16)
FILL_ITERS = 1000            # iteration cap of the diffusion fill's conjugate gradients (a map stops at tol * |r0|)


def compute_flows(raft, frames, iters=20, batch=RAFT_PAIR_BATCH, enc_batch=16):
    """frames [N,3,H,W] in 0..255 (H, W multiples of 8) -> (forward [N-1,2,H,W], backward [N-1,2,H,W]).
    forward[i] = RAFT(frame i, frame i+1), backward[i] = RAFT(frame i+1, frame i)  (tool/video_inpainting.py:246-263)."""
    N = frames.shape[0]
    with torch.no_grad():
        fm, cm = [], []
        for s in range(0, N, enc_batch):
            packed = raft.pack_images(frames[s:s + enc_batch])
            fm.append(raft.encode_features(packed))
            cm.append(raft.encode_context(packed))
        fmap, cmap = torch.cat(fm, 0), torch.cat(cm, 0)
        i1 = list(range(0, N - 1)) + list(range(1, N))          # forward pairs then backward pairs
        i2 = list(range(1, N)) + list(range(0, N - 1))
        ups = []
        for s in range(0, len(i1), batch):
            a = torch.tensor(i1[s:s + batch], device=frames.device)
            b = torch.tensor(i2[s:s + batch], device=frames.device)
            _, up = raft.iterate(fmap[a], fmap[b], cmap[a], iters=iters, test_mode=True)
            ups.append(up)
        up = torch.cat(ups, 0)
        return up[: N - 1], up[N - 1:]


def indices_gen(pivot, interval, frames, t):
    """tool/video_inpainting.py:90-100: reflect-indexed temporal neighbourhood of `pivot`."""
    out = []
    for i in range(-(frames // 2), frames // 2 + 1):
        idx = pivot + interval * i
        if idx < 0:
            idx = abs(idx)
        if idx > t - 1:
            idx = 2 * (t - 1) - idx
        out.append(idx)
    return out


def diffusion(flows, masks, iters=FILL_ITERS, tol=1e-6):
    """`diffusion()` of tool/video_inpainting.py:42-51 (rf.regionfill per flow and channel, tool/utils/region_fill.py:7-63) for the
    whole clip in one call: flows [1,2,t,H,W], masks [1,1,t,H,W] (non-zero = hole) -> diffused flows [1,2,t,H,W].
    All 2t maps are solved together on the GPU (ops.laplace_fill); map (c, i) uses mask i."""
    _, c, t, H, W = flows.shape
    out = ops.laplace_fill(flows[0].reshape(c * t, H, W).float(), masks[0, 0], iters=iters, tol=tol)
    return out.view(1, c, t, H, W)


def complete_flows(lafc, flows, masks, diffused=None, num_flows=3, interval=3, batch=LAFC_PIVOT_BATCH):
    """`complete_flow` of tool/video_inpainting.py:341-386: diffusion fill (when `diffused` is not given) + the LAFC loop of
    :367-384 with `batch` pivots per LAFC call.
    flows, diffused [1,2,t,H,W]; masks [1,1,t,H,W] (already sliced for the direction, :350-353).  Returns [t,2,H,W]:
    completed flow inside the mask, the known flow outside (`output * pivot_mask + pivot_flow * (1 - pivot_mask)`)."""
    if diffused is None:
        diffused = diffusion(flows, masks)
    t = diffused.shape[2]
    pivot = num_flows // 2
    idx = torch.tensor([indices_gen(i, interval, num_flows, t) for i in range(t)], device=flows.device)    # [t, num_flows]
    out = []
    with torch.no_grad():
        for s in range(0, t, batch):
            ii = idx[s:s + batch]                                            # [b, num_flows]
            inp = diffused[0][:, ii].permute(1, 0, 2, 3, 4)                   # [b, 2, num_flows, H, W]
            cm = masks[0][:, ii].permute(1, 0, 2, 3, 4)                       # [b, 1, num_flows, H, W]
            flow = lafc(inp.contiguous(), cm.contiguous())[0]                 # [b, 2, H, W]
            pm = cm[:, :, pivot]
            pf = flows[0][:, ii[:, pivot]].permute(1, 0, 2, 3)
            out.append(flow * pm + pf * (1 - pm))
    return torch.cat(out, 0)
