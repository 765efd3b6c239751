"""bench.py's collective pre-flight (fgt_amd/preflight.py) over gloo, world 2 and 3: the three collectives of the sharded clip through the scheduler's
own wrappers, uneven / zero-length splits included; a failing collective is named instead of raised."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q, break_a2a):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fgt_amd import preflight
    if break_a2a:
        def boom(*a, **k):
            raise RuntimeError("injected failure")
        preflight.all_to_all_rows = boom
    r = preflight.collective_preflight("cpu", rank, world, frame_floats=1000, frames_per_rank=3, u8_bytes_per_rank=5000, row_floats=64, reps=2)
    q.put((rank, r))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, port, break_a2a=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, break_a2a)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_preflight_gloo_world_3_all_three_collectives():
    out = _run(3, 29731)
    for rank, r in out.items():
        assert r["failed"] is None and r["world"] == 3
        assert list(r["checks"]) == ["all_gather_fp32_feature_chunk", "all_gather_uint8_window_outputs", "all_to_all_uneven_zero_length_splits"]
        assert all("us" in c and c["us"] > 0 for c in r["checks"].values())
    assert out[0]["checks"]["all_gather_fp32_feature_chunk"]["bytes_received_per_rank"] == 2 * 3 * 1000 * 4


def test_preflight_names_the_failing_collective():
    out = _run(2, 29733, break_a2a=True)
    for r in out.values():
        assert r["failed"] == "all_to_all_uneven_zero_length_splits" and "injected failure" in r["checks"][r["failed"]]["error"]
        assert "us" in r["checks"]["all_gather_fp32_feature_chunk"]


def test_split_plan_is_consistent_and_has_zero_lengths():
    from fgt_amd.preflight import a2a_split_plan
    for world in (2, 3, 8):
        plans = [a2a_split_plan(r, world) for r in range(world)]
        for r in range(world):
            for q in range(world):
                assert plans[r][0][q] == plans[q][1][r]          # what r sends to q is what q expects from r
        assert any(0 in p[0] for p in plans) and len({tuple(p[0]) for p in plans}) > 1
