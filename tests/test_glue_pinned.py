"""The tool's glue pinned to the REFERENCE'S OWN CODE, not to a second copy of the same text.

`oracle/reference_glue.py` cuts `indicesGen`, `get_ref_index`, `norm_flows` and the sliding-window compose loop out of
tool/video_inpainting.py (:90-117, :402-407, :710-740) with `ast`; tests/golden/make_golden_glue.py ran them and committed
the results.  Here:
  * CPU: `scheduler.window_schedule`, `flow_pipeline.indices_gen`, the oracle's `window_schedule` / `norm_flows` / `fgt_clip`
    and `scheduler.compose_torch` against those fixtures, and live against the extracted code when the reference is mounted;
  * GPU (-m gpu): the HIP compose kernel (`fgt_compose_blend`), `fgt_norm_flows`, `fgt_pack_frames`, `fgt_gather_rows`
    against the same fixtures — byte / index work, so `torch.equal`.
"""
import json
import os

import pytest
import torch

from fgt_amd import flow_pipeline
from fgt_amd.scheduler import ClipRunner, compose_torch, window_schedule
from oracle import fgt_oracle as O
from oracle import reference_glue as RG
from util import GOLDEN, load_golden

torch.set_grad_enabled(False)


def _schedules():
    return json.load(open(os.path.join(GOLDEN, "glue_schedules.json")))


def test_window_schedule_matches_reference_fixture():
    for key, rows in _schedules()["schedules"].items():
        n, stride, step, num_ref = map(int, key.split(","))
        want = [(nb, ref) for nb, ref in rows]
        assert window_schedule(n, stride, step, num_ref) == want, key
        assert O.window_schedule(n, stride, step, num_ref) == want, key


def test_indices_gen_matches_reference_fixture():
    for key, want in _schedules()["indicesGen"].items():
        p, i, fr, t = map(int, key.split(","))
        assert flow_pipeline.indices_gen(p, i, fr, t) == want, key


def test_norm_flows_oracle_matches_reference_fixture():
    g = load_golden("glue_norm_flows.npz")
    assert torch.equal(O.norm_flows(g["flows"]), g["normed"])


def _golden_clip():
    g = load_golden("glue_compose_23.npz")
    outs = [g[f"out{i}"] for i in range(len(g["log"]))]
    return g, outs


def test_compose_loop_oracle_and_cpu_runner_match_reference_fixture():
    g, outs = _golden_clip()
    it = iter(outs)
    ref = O.fgt_clip(None, None, g["frames01"], g["flows"], g["masks"], forward=lambda a, b, c: next(it))
    assert torch.equal(ref, g["comp"])
    it = iter(outs)
    got = ClipRunner(None, g["frames01"], g["flows"], g["masks"], forward=lambda a, b, c: next(it)).run()
    assert torch.equal(got, g["comp"])
    # frame 5 is composed by three windows (0, 5, 10): the order-dependent running average is exercised
    sched = window_schedule(23)
    assert sum(5 in nb for nb, _ in sched) == 3
    assert [(f, len(nb), len(ref)) for f, (nb, ref) in zip(range(0, 23, 5), sched)] == [tuple(r) for r in g["log"].tolist()]


@pytest.mark.skipif(not RG.available(), reason="reference tree not mounted (GPU box)")
def test_live_against_extracted_reference_code():
    fns = RG.functions()
    for n in (1, 4, 5, 6, 17, 46, 80, 161):
        for stride, step, num_ref in ((5, 10, -1), (5, 10, 4), (3, 7, 2), (4, 4, -1), (5, 10, 0)):
            want = []
            for f in range(0, n, stride):
                nb = list(range(max(0, f - stride), min(n, f + stride + 1)))
                want.append((nb, fns["get_ref_index"](f, nb, n, step, num_ref)))
            assert window_schedule(n, stride, step, num_ref) == want
    for t in (1, 2, 3, 7, 80):
        for p in range(t):
            for interval, frames in ((3, 3), (1, 5), (2, 3)):
                if t == 1 and frames > 1:
                    continue        # the reference itself indexes out of range for a single-flow clip
                assert flow_pipeline.indices_gen(p, interval, frames, t) == fns["indicesGen"](p, interval, frames, t)
    g = torch.Generator().manual_seed(3)
    fl = torch.randn(1, 4, 2, 9, 13, generator=g)
    assert torch.equal(O.norm_flows(fl), fns["norm_flows"](fl))
    # the whole loop with a stand-in model, another clip length / stride than the fixture
    n, H, W = 14, 8, 12
    fr = torch.rand(1, n, 3, H, W, generator=g)
    ms = (torch.rand(1, n, 1, H, W, generator=g) > 0.4).float()
    model = lambda mf, f2, m2: torch.tanh(mf[0] * 1.3 + f2[0].mean(1, keepdim=True) * 0.3 - m2[0] * 0.2)
    comp, _ = RG.window_loop()(model, fr, ms, fl.new_zeros(1, n, 2, H, W), 3, 4, -1)
    ref = torch.stack([torch.from_numpy(c).float() for c in comp], 0)
    assert torch.equal(O.fgt_clip(None, None, fr, fl.new_zeros(1, n, 2, H, W), ms, 3, 4, -1, forward=model), ref)


# ------------------------------------------------------------------------------------------------ GPU: the HIP kernels
@pytest.mark.gpu
def test_hip_compose_blend_equals_reference_loop(dev):
    """fgt_compose_blend (csrc/pointwise.hip compose_kernel) vs the reference loop's comp_frames: bit-exact, frame 5 visited 3x."""
    from fgt_amd import ops
    g, outs = _golden_clip()
    n, H, W = 23, 16, 24
    sched = window_schedule(n)
    f01, mk = g["frames01"][0].contiguous().to(dev), g["masks"][0].contiguous().to(dev)
    comp = torch.full((n, H, W, 3), -7.0, device=dev)
    seen = set()
    for (nb, _), out in zip(sched, outs):
        first = torch.tensor([0 if i in seen else 1 for i in nb], dtype=torch.int32, device=dev)
        seen.update(nb)
        ops.compose_blend(out[: len(nb)].to(dev), torch.tensor(nb, dtype=torch.int32, device=dev), first, f01, mk, comp)
    assert torch.equal(comp.cpu(), g["comp"])
    # and through ClipRunner's own GPU branch (index tensors, `first` flags and ordering built by the scheduler)
    it = iter(outs)
    got = ClipRunner(None, g["frames01"].to(dev), g["flows"].to(dev), g["masks"].to(dev), forward=lambda a, b, c: next(it).to(dev)).run()
    assert torch.equal(got.cpu(), g["comp"])


@pytest.mark.gpu
def test_hip_norm_flows_pack_gather_bit_exact(dev):
    from fgt_amd import ops
    g = load_golden("glue_norm_flows.npz")
    got = ops.norm_flows(g["flows"].to(dev))
    assert torch.equal(got.cpu(), g["normed"])
    dup = ops.norm_flows(g["flows"].to(dev), n_out=6)                  # tool/video_inpainting.py:705: last flow duplicated
    assert torch.equal(dup[0, :5].cpu(), g["normed"][0]) and torch.equal(dup[0, 5].cpu(), g["normed"][0, 4])
    big = torch.randn(1, 3, 2, 240, 432, generator=torch.Generator().manual_seed(1)) * 5
    assert torch.equal(ops.norm_flows(big.to(dev)).cpu(), O.norm_flows(big))
    c, _ = _golden_clip()
    fr, ms = c["frames01"], c["masks"]
    ids = torch.tensor([3, 0, 22, 7, 7], dtype=torch.int32)
    want = torch.cat([(fr[0, ids.long()] * 2 - 1) * (1 - ms[0, ids.long()]), ms[0, ids.long()]], 1).permute(0, 2, 3, 1)
    got = ops.pack_frames(fr[0].contiguous().to(dev), ms[0].contiguous().to(dev), ids.to(dev))
    assert torch.equal(got.cpu(), want)
    got = ops.pack_frames(fr[0].contiguous().to(dev), ms[0].contiguous().to(dev))
    assert torch.equal(got.cpu(), torch.cat([(fr[0] * 2 - 1) * (1 - ms[0]), ms[0]], 1).permute(0, 2, 3, 1))
    src = torch.randn(9, 5, 28, generator=torch.Generator().manual_seed(2))
    assert torch.equal(ops.gather_rows(src.to(dev), ids.to(dev) % 9).cpu(), src[(ids % 9).long()])
