"""Host scheduling / compose logic on CPU: ClipRunner vs the oracle's restatement of tool/video_inpainting.py:687-740,
single process and window-sharded over 2 gloo ranks (the N>1 path of bench.py)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fgt_amd.scheduler import ClipRunner, assign_windows, ideal_speedup, needs_host_staging, window_cost, window_schedule
from oracle import fgt_oracle as O

torch.set_grad_enabled(False)


def cheap_forward(mf, fl, ms):
    """Stand-in for the model: any deterministic map [1,t,3,H,W] -> [t,3,H,W] in (-1,1) that depends on all inputs."""
    x = mf[0] * 0.7 + fl[0].mean(1, keepdim=True) * 0.2 + ms[0] * 0.1
    return torch.tanh(x + 0.05 * x.mean(0, keepdim=True))


def clip(n, H=16, W=24, seed=0):
    g = torch.Generator().manual_seed(seed)
    fr = torch.rand(1, n, 3, H, W, generator=g)
    ms = (torch.rand(1, n, 1, H, W, generator=g) > 0.6).float()
    fl = O.norm_flows(torch.randn(1, n, 2, H, W, generator=g))
    return fr, fl, ms


def test_schedule_matches_oracle_and_reference_log():
    for n in (7, 20, 33, 80, 160):
        assert window_schedule(n) == O.window_schedule(n)
    assert window_schedule(40, 5, 10, 4) == O.window_schedule(40, 5, 10, 4)
    s = window_schedule(80)
    assert [len(a) + len(b) for a, b in s] == [13, 17, 18, 17, 18, 17, 18, 17, 18, 17, 18, 17, 18, 17, 18, 17]


def test_assign_windows_is_a_balanced_partition_that_co_locates_equal_lengths():
    for n in (23, 80, 160):
        sched = window_schedule(n)
        cost = lambda ws: sum(window_cost(len(sched[w][0]) + len(sched[w][1]), len(sched[w][0])) for w in ws)
        for world in (1, 2, 3, 4, 8):
            parts = assign_windows(sched, world)
            assert sorted(sum(parts, [])) == list(range(len(sched)))
            assert all(p == sorted(p) for p in parts)
            if len(sched) >= world:
                assert max(map(cost, parts)) <= 1.25 * cost(range(len(sched))) / world + max(cost([w]) for w in range(len(sched))) * (len(sched) % world != 0)
            assert abs(ideal_speedup(sched, world) - cost(range(len(sched))) / max(map(cost, parts))) < 1e-9
    # the 80-frame bench clip on 8 ranks: 4 x (17,17), 3 x (18,18), (18,13): every rank but one batches its two windows
    sched = window_schedule(80)
    t = lambda w: len(sched[w][0]) + len(sched[w][1])
    parts = assign_windows(sched, 8)
    assert sorted(sorted(t(w) for w in p) for p in parts) == sorted([[17, 17]] * 4 + [[18, 18]] * 3 + [[13, 18]])
    assert ideal_speedup(sched, 8) > 7.4
    parts = assign_windows(sched, 2)
    assert sorted(sorted(t(w) for w in p) for p in parts) == sorted([[17] * 8, [13] + [18] * 7])


def test_rccl_branch_gathers_device_buffers_directly(monkeypatch):
    """backend "nccl" (= RCCL on ROCm): all_gather_into_tensor gets the device tensors themselves; only gloo stages through the host."""
    import torch.distributed as dist
    from fgt_amd import scheduler
    assert needs_host_staging(True, "gloo") and not needs_host_staging(True, "nccl") and not needs_host_staging(False, "gloo")
    # a group created without an explicit backend reports a composite string: still RCCL for device tensors
    assert not needs_host_staging(True, "cuda:nccl,cpu:gloo") and needs_host_staging(True, "cpu:gloo")
    calls = []

    class FakeDev:                                   # stands in for a device tensor on this CPU-only box
        is_cuda = True
        shape, dtype = (2, 3), torch.float32

        def cpu(self):
            calls.append("cpu")
            return torch.zeros(2, 3)

        def copy_(self, other):
            calls.append("copy_")

    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(dist, "all_gather_into_tensor", lambda out, buf, group=None, async_op=False: calls.append(("ag", out, buf, async_op)) or "work")
    out, buf = FakeDev(), FakeDev()
    assert scheduler.all_gather(out, buf, async_op=True) == "work"
    assert calls == [("ag", out, buf, True)]         # no host staging, the handle of the asynchronous collective is returned
    calls.clear()
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "gloo")
    scheduler.all_gather(out, buf, async_op=True).wait()
    assert calls[0] == "cpu" and calls[-1] == "copy_"


@pytest.mark.parametrize("n", [6, 23, 40])
def test_cliprunner_matches_oracle_clip(n):
    fr, fl, ms = clip(n)
    ref = O.fgt_clip(None, None, fr, fl, ms, forward=cheap_forward)
    got = ClipRunner(None, fr, fl, ms, forward=cheap_forward).run()
    assert torch.equal(got, ref)          # integer-valued uint8 compose + exact 0.5/0.5 averages: bit exact


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fr, fl, ms = clip(n)
    got = ClipRunner(None, fr, fl, ms, forward=cheap_forward, rank=rank, world=world).run()
    q.put((rank, got.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [23, 40])
def test_window_sharding_two_ranks_gloo(n):
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=120) for _ in range(world))}
    [p.join(timeout=60) for p in procs]
    fr, fl, ms = clip(n)
    ref = O.fgt_clip(None, None, fr, fl, ms, forward=cheap_forward)
    for r in range(world):
        assert torch.equal(res[r], ref), f"rank {r} differs from the single-process result"


def test_window_groups_partition_the_rank_windows():
    """ClipRunner.groups: every window of the rank exactly once, equal lengths inside a group, group size bounded by window_batch and
    by the attention launch limit (_max_batch: frames x windows x heads <= 65535 grid.y entries)."""
    import torch
    from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
    from fgt_amd.scheduler import ClipRunner
    m = Model(dict(DEFAULT_CONFIG))
    for (n, H, W, wb, world) in ((80, 240, 432, 8, 1), (80, 240, 432, 4, 3), (160, 480, 864, 8, 1), (40, 1080, 1920, 8, 2)):
        fr = torch.zeros(1, n, 3, 8, 8)
        for rank in range(world):
            r = ClipRunner(m, fr, torch.zeros(1, n, 2, 8, 8), torch.zeros(1, n, 1, 8, 8), rank=rank, world=world, forward=lambda *a: None,
                           cache_features=False, window_batch=wb)
            r.H, r.W = H, W                                     # geometry of the real clip (the tensors above are placeholders)
            r.__init__(m, fr, torch.zeros(1, n, 2, 8, 8), torch.zeros(1, n, 1, 8, 8), rank=rank, world=world, forward=lambda *a: None,
                       cache_features=False, window_batch=wb)
            assert sorted(w for g in r.groups for w in g) == r.mine
            for g in r.groups:
                ts = {len(r.sched[w][0]) + len(r.sched[w][1]) for w in g}
                assert len(ts) == 1 and len(g) <= wb
    big = ClipRunner.__new__(ClipRunner)
    big.model, big.H, big.W = m, 1080, 1920
    assert big._max_batch(26) == 2 and big._max_batch(200) == 1
    small = ClipRunner.__new__(ClipRunner)
    small.model, small.H, small.W = m, 240, 432
    assert small._max_batch(17) == 64


# ------------------------------------------------------------------------------------------------ 8 ranks (the node the driver scales to)
class _StubNet:
    """The smallest model the feature-cache path of ClipRunner can drive: per-frame features that identify their frame, a window
    forward that depends on EVERY frame of the window (so a wrong or missing feature row changes the composite)."""
    passmask, in_channels = 1, 4
    cfg = dict(cnum=2, c=4, cf=2, p=(3, 3), k=(7, 7), s=(3, 3), ws=8, heads=4, flow_in=2)

    def token_grid(self, H, W):
        return tuple((n // 4 + 6 - 7) // 3 + 1 for n in (H, W))

    def encode_frames(self, masked, flows, masks, packed_in=None, out=None):
        b, t, _, H, W = masked.shape
        th, tw = self.token_grid(H, W)
        pool = torch.nn.functional.avg_pool2d
        enc = torch.cat([pool(masked[0], 4), pool(masks[0], 4)], 1).permute(0, 2, 3, 1).contiguous()             # [t, H/4, W/4, 4]
        tok = pool(masked[0], (H // th, W // tw))[:, :, :th, :tw].permute(0, 2, 3, 1)
        tok = torch.cat([tok, tok[..., :1] * 0.5], -1).reshape(t * th * tw, 4)
        ftok = pool(flows[0], (H // th, W // tw))[:, :, :th, :tw].permute(0, 2, 3, 1).reshape(t * th * tw, 2)
        return enc, tok, ftok, th, tw

    def transform_decode(self, enc, x, f, b, t, th, tw, keep=None, tq=None, keep_q=None):
        n = th * tw
        ctx = (x.view(b, t, n, 4).mean((1, 2)) + f.view(b, t, n, 2).mean((1, 2)).sum(-1, keepdim=True) * 0.3)         # [b, 4]: all t frames of a window
        k = keep.long()
        e = enc[k]                                                                                                   # [nk, Hf, Wf, 4]
        y = e[..., :3] * 0.8 + ctx[k // t][:, None, None, :3] * 0.5 + e[..., 3:] * 0.1
        return torch.tanh(torch.nn.functional.interpolate(y.permute(0, 3, 1, 2), scale_factor=4, mode="nearest"))


class _StubModel:
    net = _StubNet()


def _worker8(rank, world, port, n, H, W, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    fr, fl, ms = clip(n, H, W)
    r = ClipRunner(_StubModel(), fr, fl, ms, rank=rank, world=world, cache_features=True, encode_chunk=4)
    r.timing = True
    got = r.run()
    q.put((rank, got.numpy(), r.rows, r.n_chunks, sorted(r.phase_ms())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,what", [(80, "BASELINE C3 schedule: 16 windows"), (160, "BASELINE C5 schedule: 32 windows")])
def test_eight_ranks_needed_rows_exchange_matches_single_rank(n, what):
    """The sharded clip on 8 gloo ranks (frames block-sharded, features delivered by the needed-rows all-to-all, windows cost-balanced,
    uint8 exchange, ordered blend) == the single-rank composite on every rank, for the two schedules the driver benchmarks; a rank
    holds only the frames its windows use; the assignment stays within the bound the bench line reports."""
    world, port, H, W = 8, 33500 + (os.getpid() % 2000), 32, 48
    fr, fl, ms = clip(n, H, W)
    single = ClipRunner(_StubModel(), fr, fl, ms, cache_features=True)
    assert single.cache_features
    want = single.run()
    sched = single.sched
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, n, H, W, q)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: (torch.from_numpy(a), rows, nch, ph) for r, a, rows, nch, ph in (q.get(timeout=300) for _ in range(world))}
    [p.join(timeout=60) for p in procs]
    parts = assign_windows(sched, world)
    cost = lambda ws: sum(window_cost(len(sched[w][0]) + len(sched[w][1]), len(sched[w][0])) for w in ws)
    assert max(map(cost, parts)) <= cost(range(len(sched))) / ideal_speedup(sched, world) * (1 + 1e-9)
    assert ideal_speedup(sched, world) > (7.4 if n == 80 else 7.5)
    for r in range(world):
        got, rows, nch, phases = res[r]
        assert torch.equal(got, want), f"{what}: rank {r} differs from the single-rank composite"
        need = {f for w in parts[r] for f in sched[w][0] + sched[w][1]}
        assert rows == len(need) and rows <= 0.45 * n, f"rank {r} holds {rows} feature rows of {n} frames"
        assert nch == max(2, -(-(-(-n // world)) // 4))               # chunk count follows encode_chunk (= 4 here), at least two
        assert phases == ["blend", "encode", "exchange", "gather_wait", "windows"]


def _worker_ag(rank, world, port, n, H, W, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FGT_EXCHANGE="allgather")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    fr, fl, ms = clip(n, H, W)
    r = ClipRunner(_StubModel(), fr, fl, ms, rank=rank, world=world, cache_features=True, encode_chunk=4)
    got = r.run()
    q.put((rank, got.numpy(), r.rows, r.exchange))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(80, 8), (23, 3), (5, 8)])
def test_allgather_exchange_mode_matches_single_rank(n, world):
    """FGT_EXCHANGE=allgather (the degraded mode for a first RCCL run: plain equal-sized all_gather_into_tensor per chunk instead of the
    needed-rows all-to-all): same composite as one rank, also with ragged blocks (23 frames / 3 ranks) and ranks past the clip (5 / 8)."""
    port, H, W = 35500 + (os.getpid() % 2000), 32, 48
    fr, fl, ms = clip(n, H, W)
    want = ClipRunner(_StubModel(), fr, fl, ms, cache_features=True).run()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ag, args=(r, world, port, n, H, W, q)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: (torch.from_numpy(a), rows, ex) for r, a, rows, ex in (q.get(timeout=300) for _ in range(world))}
    [p.join(timeout=60) for p in procs]
    for r in range(world):
        got, rows, ex = res[r]
        assert ex == "allgather" and rows >= n
        assert torch.equal(got, want), f"rank {r} differs from the single-rank composite"
