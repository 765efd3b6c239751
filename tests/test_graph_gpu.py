"""hipGraph replay (fgt_amd/graph.py) must reproduce the eager launch sequence bit for bit: FGT clip runner, RAFT pair, LAFC call."""
import argparse
import json
import os

import pytest
import torch

from fgt_amd.synth import synth_clip, synth_state_dict
from util import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _sd(name, mode="kaiming"):
    keys = json.load(open(os.path.join(GOLDEN, name)))
    tmpl = {k: torch.empty(v, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32) for k, v in keys.items()}
    return synth_state_dict(tmpl, seed=0, mode=mode)


def test_cliprunner_graph_replay_equals_eager(dev):
    from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
    from fgt_amd.scheduler import ClipRunner
    m = Model(dict(DEFAULT_CONFIG)).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
    m = m.to(dev)
    fr, fl, ms = synth_clip(23, 64, 96, seed=3, device=dev)
    eager = ClipRunner(m, fr, fl, ms, use_graphs=False).run()
    r = ClipRunner(m, fr, fl, ms, use_graphs=True)
    a, b = r.run(), r.run()                       # capture pass, then pure replay
    assert torch.equal(a, eager) and torch.equal(b, eager)


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_window_batching_bit_equal(prec, dev, monkeypatch):
    """Batching equal-length windows as b > 1 (ClipRunner.window_batch) changes launch sizes only: bit-identical composite,
    eager and under hipGraph replay, in both arithmetic modes (bf16x3 also walks the pre-split conv chains)."""
    from fgt_amd import ops
    from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
    from fgt_amd.scheduler import ClipRunner
    monkeypatch.setattr(ops, "DEFAULT_CONV_PRECISION", prec)
    monkeypatch.setattr(ops, "DEFAULT_ATTN_PRECISION", prec)
    m = Model(dict(DEFAULT_CONFIG)).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
    m = m.to(dev)
    fr, fl, ms = synth_clip(46, 64, 96, seed=5, device=dev)
    one = ClipRunner(m, fr, fl, ms, use_graphs=False, window_batch=1)
    four = ClipRunner(m, fr, fl, ms, use_graphs=False, window_batch=4)
    assert max(len(g) for g in four.groups) >= 3
    a = one.run()
    assert torch.equal(four.run(), a)
    g = ClipRunner(m, fr, fl, ms, use_graphs=True, window_batch=3)
    assert torch.equal(g.run(), a) and torch.equal(g.run(), a)


def test_raft_and_lafc_graph_replay_equal_eager(dev):
    from fgt_amd import lafc_model, raft_model
    r = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    r.load_state_dict(_sd("raft_state_keys.json"), strict=True)
    r = r.to(dev)
    g = load_golden("raft_128x160_it6.npz")
    i1, i2 = g["image1"].to(dev), g["image2"].to(dev)
    lo, up = r(i1, i2, iters=6, test_mode=True)
    r.use_graph = True
    lo2, up2 = r(i1, i2, iters=6, test_mode=True)
    lo3, up3 = r(i2, i1, iters=6, test_mode=True)           # replay with other inputs ...
    lo4, up4 = r(i1, i2, iters=6, test_mode=True)           # ... and back
    assert torch.equal(lo, lo2) and torch.equal(up, up2) and torch.equal(lo, lo4) and torch.equal(up, up4)
    assert not torch.equal(up3, up)
    m = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG)).eval()
    m.load_state_dict(_sd("lafc_vanilla_state_keys.json"), strict=True)
    m = m.to(dev)
    gl = load_golden("lafc_vanilla_64x96.npz")
    a = m(gl["flows"].to(dev), gl["masks"].to(dev))
    m.net.use_graph = True
    b = m(gl["flows"].to(dev), gl["masks"].to(dev))
    c = m(gl["flows"].to(dev), gl["masks"].to(dev))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[0], c[0])
