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
