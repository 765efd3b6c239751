"""Rank sharding of the flow stages (RAFT pairs, fill maps, LAFC pivots and blend frames are
independent units, tool/video_inpainting.py:246-263, 342-385, 644-682) on CPU: gloo world 3 (ragged blocks) and world 8 on a 4-frame clip
(ranks past the clip contribute empty blocks) over the CPU kernel spec (tests/fake_ops.py) must reproduce the single-rank stages.
The HIP-vs-HIP `torch.equal` form of this test runs on the GPU box (tests/test_dist_gpu.py)."""
import argparse
import json
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fake_ops
from util import GOLDEN

torch.set_grad_enabled(False)


def _sd(name, mode="kaiming"):
    from fgt_amd.synth import synth_state_dict
    keys = json.load(open(os.path.join(GOLDEN, name)))
    tmpl = {k: torch.empty(v, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32) for k, v in keys.items()}
    return synth_state_dict(tmpl, seed=0, mode=mode)


def _setup():
    from fgt_amd import flow_pipeline, lafc_model, raft_model
    for m in (lafc_model, raft_model, flow_pipeline):
        m.ops = fake_ops
    for m in (lafc_model, raft_model):
        m.PackedConv = fake_ops.PackedConv
    raft = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    raft.load_state_dict(_sd("raft_state_keys.json"), strict=True)
    lafc = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
    lafc.load_state_dict(_sd("lafc_vanilla_state_keys.json"), strict=True)
    return flow_pipeline, raft, lafc


def _inputs(N, H=128, W=160):
    g = torch.Generator().manual_seed(11)
    frames = torch.rand(N, 3, H, W, generator=g) * 255.0
    masks = torch.zeros(1, 1, N - 1, H, W)
    for i in range(N - 1):
        masks[0, 0, i, 30 + i:70 + i, 40:100] = 1
    return frames, masks


def _stages(fp, raft, lafc, frames, masks, rank=0, world=1):
    kw = dict(rank=rank, world=world)
    fw, bw = fp.compute_flows(raft, frames, iters=2, batch=2, enc_batch=2, **kw)
    flows = fw.permute(1, 0, 2, 3)[None].contiguous()
    dif = fp.diffusion(flows, masks, **kw)
    comp = fp.complete_flows(lafc, flows, masks, dif, batch=2, **kw)
    return fw, bw, dif, comp


def _worker(rank, world, port, N, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    fp, raft, lafc = _setup()
    frames, masks = _inputs(N)
    out = _stages(fp, raft, lafc, frames, masks, rank, world)
    q.put((rank, [o.numpy() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_and_batches():
    from fgt_amd import flow_pipeline as fp
    for n in (1, 5, 79, 80, 159):
        for world in (1, 2, 3, 8):
            blocks = [fp.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            per = -(-n // world)
            assert all(hi - lo == per for lo, hi in blocks if hi < n)         # only the tail is short: gather_blocks' [:n] trim is exact
    # the batch follows a memory budget: 64 pairs at 864x480 (14.3 GB of pyramids), a handful at 1080p, never zero
    assert fp.raft_pair_batch(480, 864) == 64 and fp.raft_pair_batch(240, 432) == 64
    assert 1 <= fp.raft_pair_batch(1080, 1920) <= 8 and fp.raft_pair_batch(2160, 3840) == 1
    assert fp.lafc_pivot_batch(240, 432) == 16 and 1 <= fp.lafc_pivot_batch(1080, 1920) <= 4


@pytest.mark.parametrize("N,world", [(6, 3), (4, 8)])
def test_sharded_flow_stages_match_single_rank(N, world):
    port = 37500 + (os.getpid() % 2000)
    fp, raft, lafc = _setup()
    frames, masks = _inputs(N)
    want = _stages(fp, raft, lafc, frames, masks)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: [torch.from_numpy(a) for a in outs] for r, outs in (q.get(timeout=600) for _ in range(world))}
    [p.join(timeout=60) for p in procs]
    names = ("forward flows", "backward flows", "diffused flows", "completed flows")
    for r in range(world):
        for name, got, ref in zip(names, res[r], want):
            assert got.shape == ref.shape, (name, got.shape, ref.shape)
            # CPU spec: torch's conv kernels may pick another algorithm for another batch size; the HIP path is bit-identical (GPU test)
            err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
            assert err < 2e-5, f"rank {r} {name}: rel err {err}"
        for a, b in zip(res[r], res[0]):
            assert torch.equal(a, b)                                           # every rank holds the same gathered result
