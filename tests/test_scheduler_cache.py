"""Per-frame feature cache + frame-sharded encode (exact dedup of the per-frame stages) on CPU over the kernel spec:
cached ClipRunner == uncached ClipRunner == oracle clip, single process and 2 gloo ranks."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fake_ops
from fgt_amd import fgt_model
from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
from fgt_amd.scheduler import ClipRunner
from fgt_amd.synth import synth_state_dict
from oracle import fgt_oracle as O

torch.set_grad_enabled(False)
N, H, W = 12, 32, 48


def _setup(N=N):
    fgt_model.ops = fake_ops
    fgt_model.PackedConv = fake_ops.PackedConv
    m = Model(dict(DEFAULT_CONFIG)).eval()
    sd = synth_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(0)
    fr = torch.rand(1, N, 3, H, W, generator=g)
    ms = (torch.rand(1, N, 1, H, W, generator=g) > 0.6).float()
    fl = O.norm_flows(torch.randn(1, N, 2, H, W, generator=g))
    return m, sd, fr, fl, ms


def test_cached_equals_uncached_equals_oracle(monkeypatch):
    real_ops, real_pc = fgt_model.ops, fgt_model.PackedConv
    try:
        m, sd, fr, fl, ms = _setup()
        a = ClipRunner(m, fr, fl, ms, cache_features=False).run()
        b = ClipRunner(m, fr, fl, ms, cache_features=True, encode_chunk=5).run()
        ref = O.fgt_clip(sd, DEFAULT_CONFIG, fr, fl, ms)
        # uint8 truncation makes the composed clip piecewise constant: allow rare +-1 flips from 1e-7 round-off
        assert (a - ref).abs().max().item() <= 1.0 and ((a - ref).abs() > 0).float().mean().item() < 1e-3
        assert (b - a).abs().max().item() <= 1.0 and ((b - a).abs() > 0).float().mean().item() < 1e-3
    finally:
        fgt_model.ops, fgt_model.PackedConv = real_ops, real_pc


def test_f16_mode_clip_over_its_cpu_model(monkeypatch):
    """The 'f16' arithmetic mode through the whole clip scheduler (cache, batched windows, pruned last pair) over its CPU model
    (fake_ops rounds every Split and the weights of the GEMMs that consume one to fp16): the composite stays within one uint8 step of
    the oracle clip with a flip rate that matches an error of ~1e-4 * 127.5 steps (tests/test_f16_gpu.py holds the kernels to the same)."""
    real_ops, real_pc = fgt_model.ops, fgt_model.PackedConv
    try:
        m, sd, fr, fl, ms = _setup()
        monkeypatch.setattr(fake_ops, "DEFAULT_CONV_PRECISION", "f16")
        got = ClipRunner(m, fr, fl, ms, cache_features=True, encode_chunk=5, window_batch=4).run()
        ref = O.fgt_clip(sd, DEFAULT_CONFIG, fr, fl, ms)
        d = (got - ref).abs()
        rate = (d > 0).float().mean().item()
        print(f"[parity] f16 mode (CPU model) clip {N}x{H}x{W}: max {d.max().item()} uint8 steps, differing values {rate:.3e}")
        assert d.max().item() <= 1.0 and 0 < rate < 2e-2
    finally:
        fgt_model.ops, fgt_model.PackedConv = real_ops, real_pc


def test_last_pair_pruning_is_exact():
    """The last temporal + spatial block computed only for the frames the tool consumes (transform_decode tq / keep_q) == all frames."""
    real_ops, real_pc = fgt_model.ops, fgt_model.PackedConv
    try:
        m, sd, fr, fl, ms = _setup()
        full = ClipRunner(m, fr, fl, ms, neighbor_stride=3, ref_length=4, cache_features=True, window_batch=4, prune_last=False)
        pruned = ClipRunner(m, fr, fl, ms, neighbor_stride=3, ref_length=4, cache_features=True, window_batch=4)
        assert all(tq is None for tq in full._group_tq) and any(tq is not None for tq in pruned._group_tq)
        for g, tq, kq, k in zip(pruned.groups, pruned._group_tq, pruned._group_keep_q, pruned._group_keep):
            t = len(pruned.sched[g[0]][0]) + len(pruned.sched[g[0]][1])
            assert tq == max(len(pruned.sched[w][0]) for w in g) and kq.tolist() == [(i // t) * tq + i % t for i in k.tolist()]
        a, b = full.run(), pruned.run()
        assert (b - a).abs().max().item() <= 1.0 and ((b - a).abs() > 0).float().mean().item() < 1e-3
    finally:
        fgt_model.ops, fgt_model.PackedConv = real_ops, real_pc


def test_window_batching_is_exact():
    """Equal-length windows batched as b > 1 through transform_decode(keep=...) == one window per forward (the CPU spec's
    torch convs may differ in the last bit with the batch size, hence the same +-1 uint8 allowance as above)."""
    real_ops, real_pc = fgt_model.ops, fgt_model.PackedConv
    try:
        m, sd, fr, fl, ms = _setup()
        one = ClipRunner(m, fr, fl, ms, neighbor_stride=3, ref_length=4, cache_features=True, window_batch=1)
        four = ClipRunner(m, fr, fl, ms, neighbor_stride=3, ref_length=4, cache_features=True, window_batch=4)
        assert all(len(g) == 1 for g in one.groups) and max(len(g) for g in four.groups) >= 2
        assert sorted(w for g in four.groups for w in g) == list(range(len(four.sched)))
        a, b = one.run(), four.run()
        assert (b - a).abs().max().item() <= 1.0 and ((b - a).abs() > 0).float().mean().item() < 1e-3
    finally:
        fgt_model.ops, fgt_model.PackedConv = real_ops, real_pc


def _worker(rank, world, port, q, n=N):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, sd, fr, fl, ms = _setup(n)
    q.put((rank, ClipRunner(m, fr, fl, ms, rank=rank, world=world, cache_features=True).run().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_encode_and_window_sharding_two_ranks():
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=300) for _ in range(world))}
    [p.join(timeout=60) for p in procs]
    real_ops, real_pc = fgt_model.ops, fgt_model.PackedConv
    try:
        m, sd, fr, fl, ms = _setup()
        single = ClipRunner(m, fr, fl, ms, cache_features=True).run()
    finally:
        fgt_model.ops, fgt_model.PackedConv = real_ops, real_pc
    assert torch.equal(res[0], res[1])
    assert (res[0] - single).abs().max().item() <= 1.0 and ((res[0] - single).abs() > 0).float().mean().item() < 1e-3


def test_rank_without_frames_or_windows_keeps_the_collectives_matched():
    """4 frames on 3 ranks: blocks of 2 frames, rank 2 has no frame and ranks 1-2 no window (one window in the clip).  Every rank
    still enters every all-gather (zero rows / empty slots) and ends with the same composite — no rank-dependent raise that
    would leave the others hanging in a collective."""
    world, port, n = 3, 32500 + (os.getpid() % 2000), 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=300) for _ in range(world))}
    [p.join(timeout=60) for p in procs]
    real_ops, real_pc = fgt_model.ops, fgt_model.PackedConv
    try:
        m, sd, fr, fl, ms = _setup(n)
        single = ClipRunner(m, fr, fl, ms, cache_features=True).run()
    finally:
        fgt_model.ops, fgt_model.PackedConv = real_ops, real_pc
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
    assert (res[0] - single).abs().max().item() <= 1.0 and ((res[0] - single).abs() > 0).float().mean().item() < 1e-3
