"""Rehearsal of the multi-rank FGT stage on ONE MI355X: two ranks share cuda:0 (gloo, collectives staged through the
host), each runs the HIP model on its shard of frames / windows; the result must equal the single-rank clip.
(RCCL itself cannot be exercised on a 1-GPU box: the collective calls are the same `all_gather_into_tensor`.)"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N, H, W = 14, 64, 96


def _clip_and_model():
    from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
    from fgt_amd.synth import synth_clip, synth_state_dict
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    m = Model(dict(DEFAULT_CONFIG)).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
    fr, fl, ms = synth_clip(N, H, W, seed=5, device=dev)
    return m.to(dev), fr, fl, ms


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fgt_amd.scheduler import ClipRunner
    m, fr, fl, ms = _clip_and_model()
    comp = ClipRunner(m, fr, fl, ms, rank=rank, world=world).run()
    q.put((rank, comp.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_rank():
    from fgt_amd.scheduler import ClipRunner
    world, port = 2, 33500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=600) for _ in range(world))}
    [p.join(timeout=120) for p in procs]
    m, fr, fl, ms = _clip_and_model()
    single = ClipRunner(m, fr, fl, ms).run().cpu()
    uncached = ClipRunner(m, fr, fl, ms, cache_features=False).run().cpu()
    assert torch.equal(res[0], res[1])
    for name, other in (("2 ranks", res[0]), ("no feature cache", uncached)):
        d = (other - single).abs()
        print(f"[parity] clip {name} vs single rank: max diff {d.max().item()} (uint8 steps), differing px {(d > 0).float().mean().item():.2e}")
        assert torch.equal(other, single)      # HIP vs HIP: per-row results do not depend on batch / sharding (oracle parity: test_clip_gpu.py)


# ------------------------------------------------------------------------------------------------ flow stages sharded over ranks
FN, FH, FW = 7, 128, 160


def _flow_models(dev):
    import argparse
    import json
    from fgt_amd import lafc_model, raft_model
    from fgt_amd.synth import synth_state_dict
    from util import GOLDEN

    def sd(name):
        keys = json.load(open(os.path.join(GOLDEN, name)))
        tmpl = {k: torch.empty(v, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32) for k, v in keys.items()}
        return synth_state_dict(tmpl, seed=0, mode="kaiming")
    raft = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    raft.load_state_dict(sd("raft_state_keys.json"), strict=True)
    lafc = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
    lafc.load_state_dict(sd("lafc_vanilla_state_keys.json"), strict=True)
    return raft.to(dev), lafc.to(dev)


def _flow_stages(rank, world):
    from fgt_amd import blending, flow_pipeline, ops
    torch.set_grad_enabled(False)
    saved = (ops.DEFAULT_CONV_PRECISION, ops.DEFAULT_ATTN_PRECISION)
    ops.DEFAULT_CONV_PRECISION = ops.DEFAULT_ATTN_PRECISION = "bf16x3"
    try:
        return _flow_stages_body(rank, world)
    finally:                      # (also runs in the pytest process for the single-rank reference: the mode must not leak into later tests)
        ops.DEFAULT_CONV_PRECISION, ops.DEFAULT_ATTN_PRECISION = saved


def _flow_stages_body(rank, world):
    from fgt_amd import blending, flow_pipeline, ops
    dev = torch.device("cuda:0")
    raft, lafc = _flow_models(dev)
    g = torch.Generator().manual_seed(31)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, FH + 16, FW + 2 * FN + 16, generator=g), 7, 1, 3)
    frames = torch.cat([base[:, :, 8:8 + FH, 8 + 2 * i:8 + 2 * i + FW] for i in range(FN)], 0).contiguous().to(dev) * 255.0
    kw = dict(rank=rank, world=world)
    fw, bw = flow_pipeline.compute_flows(raft, frames, iters=3, batch=4, enc_batch=3, **kw)
    flows = fw.permute(1, 0, 2, 3)[None].contiguous()
    masks = torch.zeros(1, 1, FN - 1, FH, FW, device=dev)
    for i in range(FN - 1):
        masks[0, 0, i, 30 + i:70 + i, 40 + i:100 + i] = 1
    bounds = ops.hole_bounds(masks[0, 0])
    dif = flow_pipeline.diffusion(flows, masks, bounds=bounds, **kw)
    comp = flow_pipeline.complete_flows(lafc, flows, masks, dif, batch=2, **kw)
    img = (frames / 255.0).permute(0, 2, 3, 1).contiguous()
    hole = torch.zeros(FN, FH, FW, dtype=torch.bool, device=dev)
    hole[:, 40:80, 50:110] = True
    gx, gy = torch.zeros_like(img), torch.zeros_like(img)
    gx[:, :, :-1] = img[:, :, 1:] - img[:, :, :-1]
    gy[:, :-1] = img[:, 1:] - img[:, :-1]
    blend, unf = blending.poisson_blend_clip(img * (~hole)[..., None], gx, gy, hole, torch.zeros_like(hole), bounds=ops.hole_bounds(hole), **kw)
    return [t.float().cpu() for t in (fw, bw, dif, comp, blend, unf)]


def _flow_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _flow_stages(rank, world)
    q.put((rank, [o.numpy() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_flow_stages_match_single_rank():
    """RAFT pairs, fill maps, LAFC pivots and Poisson frames block-sharded over 2 ranks (both on cuda:0, gloo staging) with one all-gather
    per stage == the single-rank stages, bit for bit (tool/video_inpainting.py:246-263, 342-385, 644-682)."""
    world, port = 2, 35500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flow_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: [torch.from_numpy(a) for a in outs] for r, outs in (q.get(timeout=600) for _ in range(world))}
    [p.join(timeout=120) for p in procs]
    want = _flow_stages(0, 1)
    for name, a, b, w in zip(("forward flows", "backward flows", "diffused", "completed", "blend", "unfilled"), res[0], res[1], want):
        assert torch.isfinite(w).all() and torch.equal(a, b), name
        assert torch.equal(a, w), f"{name}: 2 ranks differ from 1 rank (max {float((a - w).abs().max())})"
    print("[parity] flow stages (RAFT, fill, LAFC, Poisson) on 2 ranks == 1 rank: identical")
