"""Collective pre-flight for `bench.py --gpus N` (and any first run on a new node): the three collectives the sharded clip uses, at the sizes it uses
them, through the SAME wrappers (`scheduler.all_gather`, `scheduler.all_to_all_rows`) — so that a first RCCL run that goes wrong says WHICH collective
failed or hung, and a slow one shows up as microseconds and GB/s instead of as a slow step.

  1. `all_gather_into_tensor`, fp32: one encode chunk's feature rows per rank (scheduler.ClipRunner.encode_clip, exchange = "allgather")
  2. `all_gather_into_tensor`, uint8: the window outputs (ClipRunner.exchange_outputs)
  3. `all_to_all_single` with UNEVEN and ZERO-LENGTH splits: the needed-rows exchange (exchange = "a2a") — rank r sends (r + q) % 3 rows to rank q,
     so every rank has zero-length sends and receives among its peers
Each check: 1 warm-up + `reps` timed calls, device-synchronised; its own watchdog (`timeout_s` per call): on expiry `on_hang(name, results)` is
called from the watchdog thread (bench.py emits its line with what was measured so far and exits) — a hung collective cannot be cancelled.
The result of every check is validated (rank q's block must hold rank q's pattern): a collective that returns wrong bytes is reported as an error."""
import threading
import time

import torch

from .scheduler import all_gather, all_to_all_rows


def _sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def a2a_split_plan(rank, world):
    """Rows rank `rank` sends to every peer q / receives from every peer r: (rank + q) % 3 — uneven, with zeros, consistent across ranks."""
    return [(rank + q) % 3 for q in range(world)], [(r + rank) % 3 for r in range(world)]


def collective_preflight(device, rank, world, group=None, frame_floats=60 * 108 * 128, frames_per_rank=10, u8_bytes_per_rank=2 * 6 * 3 * 240 * 432,
                         row_floats=720 * 512, reps=3, timeout_s=10.0, on_hang=None):
    """-> {"checks": {name: {"us": median per call, "GBps": bytes received per rank / time, "bytes": ...} | {"error": ...}}, "failed": name | None,
    "world": N}.  Every rank returns its own figures (rank 0's go into the bench line)."""
    device = torch.device(device)
    res = {"world": world, "reps": reps, "checks": {}, "failed": None}
    if world <= 1:
        return res

    def ag_f32():
        send = torch.full((frames_per_rank, frame_floats), float(rank + 1), dtype=torch.float32, device=device)
        recv = torch.empty((world * frames_per_rank, frame_floats), dtype=torch.float32, device=device)

        def call():
            all_gather(recv, send, group).wait()

        def check():
            got = recv.view(world, -1)[:, 0].cpu()
            assert torch.equal(got, torch.arange(1, world + 1, dtype=torch.float32)), f"wrong bytes: {got.tolist()}"
        return call, check, (world - 1) * send.numel() * 4

    def ag_u8():
        send = torch.full((u8_bytes_per_rank,), rank + 1, dtype=torch.uint8, device=device)
        recv = torch.empty((world * u8_bytes_per_rank,), dtype=torch.uint8, device=device)

        def call():
            all_gather(recv, send, group).wait()

        def check():
            got = recv.view(world, -1)[:, -1].cpu()
            assert got.tolist() == list(range(1, world + 1)), f"wrong bytes: {got.tolist()}"
        return call, check, (world - 1) * send.numel()

    def a2a():
        in_rows, out_rows = a2a_split_plan(rank, world)
        send = torch.full((max(1, sum(in_rows)), row_floats), float(rank + 1), dtype=torch.float32, device=device)
        recv = torch.zeros((max(1, sum(out_rows)), row_floats), dtype=torch.float32, device=device)

        def call():
            all_to_all_rows(recv[:sum(out_rows)], send[:sum(in_rows)], out_rows, in_rows, group).wait()

        def check():
            want = [float(r + 1) for r in range(world) for _ in range(out_rows[r])]
            got = recv[:sum(out_rows), 0].cpu().tolist()
            assert got == want, f"wrong rows: {got} != {want}"
        return call, check, sum(o for r, o in enumerate(out_rows) if r != rank) * row_floats * 4

    for name, make in (("all_gather_fp32_feature_chunk", ag_f32), ("all_gather_uint8_window_outputs", ag_u8), ("all_to_all_uneven_zero_length_splits", a2a)):
        dog = None
        if on_hang is not None:
            dog = threading.Timer(timeout_s * (reps + 1), on_hang, args=(name, res))
            dog.daemon = True
            dog.start()
        try:
            call, check, nbytes = make()
            call()
            _sync(device)
            check()
            ts = []
            for _ in range(reps):
                _sync(device)
                t0 = time.perf_counter()
                call()
                _sync(device)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            med = ts[len(ts) // 2]
            res["checks"][name] = {"us": round(med * 1e6, 1), "GBps_received_per_rank": round(nbytes / med / 1e9, 2), "bytes_received_per_rank": int(nbytes)}
        except Exception as e:  # noqa: BLE001
            res["checks"][name] = {"error": f"{type(e).__name__}: {e}"[:200]}
            res["failed"] = name
        finally:
            if dog is not None:
                dog.cancel()
        if res["failed"]:
            break
    return res
