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
from .scheduler import all_gather


RAFT_PAIR_BATCH = 64         # pairs per RAFT refinement batch (forward and backward pairs mixed): the per-iteration convs have 1 620 (432x240) /
#                              6 480 (864x480) output pixels per pair — 8 pairs left 1.6 rounds of tiles on 256 CUs.  Measured (profiles/r03_run2_raft_batch.txt):
#                              8 -> 32 pairs: 1.88 -> 1.16 ms per pair at 432x240, 4.34 -> 3.93 at 864x480, bit-identical flows, 10.6 GB peak;
#                              32 -> 64 (end of round 3, tap-reusing convs): 1.02 -> 0.95 and 3.35 -> 3.27, bit-identical, 22 GB peak
LAFC_PIVOT_BATCH = 16        # pivots per LAFC call (tools/lafc_batch.py: 8 -> 0.718, 16 -> 0.69, 32 -> 0.688 ms per flow at 432x240, bit-equal; 2.9 GB peak at 16)
FILL_ITERS = 1000            # iteration cap of the diffusion fill's conjugate gradients (a map stops at tol * |r0|)
RAFT_VOLUME_BUDGET = 24 << 30    # bytes the correlation pyramids of one refinement batch may take (288 GB per GPU; 64 pairs at 864x480 = 14.3 GB)
LAFC_BUDGET = 8 << 30            # bytes of LAFC activations per call (16 pivots at 432x240 = 2.9 GB)


def raft_pair_batch(H, W, budget=RAFT_VOLUME_BUDGET, cap=RAFT_PAIR_BATCH):
    """Pairs per RAFT refinement batch at resolution H x W: the all-pairs correlation pyramid takes 4/3 * 4 B * (H/8 * W/8)^2 per pair
    (223 MB at 864x480, 5.6 GB at 1920x1080), so the batch follows a memory budget instead of being resolution-independent (ADVICE r3:
    64 pairs at 1080p would need > 300 GB while the reference's one-pair loop runs).  Flows are bit-identical across batch sizes."""
    per_pair = (4.0 / 3.0) * 4.0 * float((H // 8) * (W // 8)) ** 2
    return int(max(1, min(cap, budget // max(per_pair, 1.0))))


def lafc_pivot_batch(H, W, budget=LAFC_BUDGET, cap=LAFC_PIVOT_BATCH):
    """Pivots per LAFC call: activations scale with H * W (181 MB per pivot at 432x240)."""
    per_pivot = 181e6 * (H * W) / (240.0 * 432.0)
    return int(max(1, min(cap, budget // per_pivot)))


# ---- rank sharding of the flow stages (RAFT pairs, LAFC flows, fill maps and blend frames are independent units,
# tool/video_inpainting.py:246-263, 369-383).  Units are block-sharded, every rank computes its block with the same kernels and batch-
# independent arithmetic, and ONE all-gather per stage hands every rank the full result (the next stage reads neighbours across block
# borders: LAFC's temporal window, propagation's flow chain).  No other collective; world == 1 is the unsharded code path.
def shard_range(n, rank, world):
    """Block [lo, hi) of n units for `rank`: blocks of ceil(n / world), later ranks may be short or empty."""
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def gather_blocks(local, n, rank, world, group=None):
    """local = this rank's block [hi - lo, ...] of an n-row tensor sharded by shard_range -> the full [n, ...] tensor on every rank.
    all_gather_into_tensor of equal-sized (zero-padded) blocks: RCCL over xGMI on the GPU box, gloo / host staging in the tests."""
    if world == 1:
        return local
    per = -(-n // world)
    send = local
    if local.shape[0] != per:
        send = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    all_gather(out, send.contiguous(), group).wait()
    return out[:n]


def compute_flows(raft, frames, iters=20, batch=None, enc_batch=16, rank=0, world=1, group=None):
    """frames [N,3,H,W] in 0..255 (H, W multiples of 8) -> (forward [N-1,2,H,W], backward [N-1,2,H,W]).
    forward[i] = RAFT(frame i, frame i+1), backward[i] = RAFT(frame i+1, frame i)  (tool/video_inpainting.py:246-263).
    world > 1: pair indices i are block-sharded; a rank encodes frames lo..hi of its block (one frame of overlap with its neighbour
    instead of a feature exchange), refines its forward and backward pairs, and one all-gather delivers all 2(N-1) fields to every rank."""
    N, _, H, W = frames.shape
    if batch is None:
        batch = raft_pair_batch(H, W)
    lo, hi = shard_range(N - 1, rank, world)
    with torch.no_grad():
        cnt = hi - lo
        if cnt > 0:
            fm, cm = [], []
            for s in range(lo, hi + 1, enc_batch):
                packed = raft.pack_images(frames[s:min(hi + 1, s + enc_batch)])
                fm.append(raft.encode_features(packed))
                cm.append(raft.encode_context(packed))
            fmap, cmap = torch.cat(fm, 0), torch.cat(cm, 0)                 # row j = frame lo + j
            i1 = list(range(0, cnt)) + list(range(1, cnt + 1))              # forward pairs then backward pairs (local frame rows)
            i2 = list(range(1, cnt + 1)) + list(range(0, cnt))
            ups = []
            for s in range(0, len(i1), batch):
                a = torch.tensor(i1[s:s + batch], device=frames.device)
                b = torch.tensor(i2[s:s + batch], device=frames.device)
                _, up = raft.iterate(fmap[a], fmap[b], cmap[a], iters=iters, test_mode=True)
                ups.append(up)
            up = torch.cat(ups, 0)
            if world == 1:
                return up[:cnt], up[cnt:]                                    # one rank: contiguous slices, no interleaving copy
            both = torch.stack([up[:cnt], up[cnt:]], 1)                      # [cnt, 2 (fwd, bwd), 2, H, W]
        else:
            both = torch.zeros(0, 2, 2, H, W, dtype=torch.float32, device=frames.device)
        both = gather_blocks(both, N - 1, rank, world, group)
        return both[:, 0], both[:, 1]


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


def diffusion(flows, masks, iters=FILL_ITERS, tol=1e-6, rank=0, world=1, group=None, bounds=None):
    """`diffusion()` of tool/video_inpainting.py:42-51 (rf.regionfill per flow and channel, tool/utils/region_fill.py:7-63) for the
    whole clip in one call: flows [1,2,t,H,W], masks [1,1,t,H,W] (non-zero = hole) -> diffused flows [1,2,t,H,W].
    All 2t maps are solved together on the GPU (ops.laplace_fill); map (c, i) uses mask i.  world > 1: the t flows are block-sharded
    (a rank solves both channels of its flows), one all-gather.  `bounds` = ops.hole_bounds(masks) of the clip (no host sync here)."""
    _, c, t, H, W = flows.shape
    if world == 1:
        out = ops.laplace_fill(flows[0].reshape(c * t, H, W).float(), masks[0, 0], iters=iters, tol=tol, bounds=bounds)
        return out.view(1, c, t, H, W)
    lo, hi = shard_range(t, rank, world)
    if hi > lo:
        loc = ops.laplace_fill(flows[0][:, lo:hi].reshape(c * (hi - lo), H, W).float(), masks[0, 0, lo:hi], iters=iters, tol=tol, bounds=bounds)
        loc = loc.view(c, hi - lo, H, W).permute(1, 0, 2, 3).contiguous()            # [cnt, c, H, W]
    else:
        loc = torch.zeros(0, c, H, W, dtype=torch.float32, device=flows.device)
    full = gather_blocks(loc, t, rank, world, group)                                    # [t, c, H, W]
    return full.permute(1, 0, 2, 3).contiguous().view(1, c, t, H, W)


def complete_flows(lafc, flows, masks, diffused=None, num_flows=3, interval=3, batch=None, rank=0, world=1, group=None):
    """`complete_flow` of tool/video_inpainting.py:341-386: diffusion fill (when `diffused` is not given) + the LAFC loop of
    :367-384 with `batch` pivots per LAFC call.
    flows, diffused [1,2,t,H,W]; masks [1,1,t,H,W] (already sliced for the direction, :350-353).  Returns [t,2,H,W]:
    completed flow inside the mask, the known flow outside (`output * pivot_mask + pivot_flow * (1 - pivot_mask)`).
    world > 1: pivots block-sharded (every rank holds the whole diffused clip: a pivot's temporal neighbours cross block borders), one
    all-gather of the completed flows."""
    if diffused is None:
        diffused = diffusion(flows, masks, rank=rank, world=world, group=group)
    t = diffused.shape[2]
    if batch is None:
        batch = lafc_pivot_batch(*flows.shape[-2:])
    lo, hi = shard_range(t, rank, world)
    pivot = num_flows // 2
    idx = torch.tensor([indices_gen(i, interval, num_flows, t) for i in range(t)], device=flows.device)    # [t, num_flows]
    out = []
    with torch.no_grad():
        for s in range(lo, hi, batch):
            ii = idx[s:min(hi, s + batch)]                                   # [b, num_flows]
            inp = diffused[0][:, ii].permute(1, 0, 2, 3, 4)                   # [b, 2, num_flows, H, W]
            cm = masks[0][:, ii].permute(1, 0, 2, 3, 4)                       # [b, 1, num_flows, H, W]
            flow = lafc(inp.contiguous(), cm.contiguous())[0]                 # [b, 2, H, W]
            pm = cm[:, :, pivot]
            pf = flows[0][:, ii[:, pivot]].permute(1, 0, 2, 3)
            out.append(flow * pm + pf * (1 - pm))
    H, W = flows.shape[-2:]
    loc = torch.cat(out, 0) if out else torch.zeros(0, 2, H, W, dtype=torch.float32, device=flows.device)
    return gather_blocks(loc, t, rank, world, group)
