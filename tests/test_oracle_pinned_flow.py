"""Pin the LAFC / RAFT / warp oracles and the host logic of their nn.Module mirrors (CPU only):
golden vectors from the reference always; the live reference when /root/reference is present."""
import argparse
import json
import os

import pytest
import torch

import fake_ops
from fgt_amd import lafc_model, raft_model
from fgt_amd.synth import synth_state_dict
from oracle import lafc_oracle as LO
from oracle import raft_oracle as RO
from oracle import reference_loader as RL
from util import GOLDEN, load_golden, max_err, rel_err

torch.set_grad_enabled(False)


def _sd(name, mode="kaiming"):
    keys = json.load(open(os.path.join(GOLDEN, name)))
    tmpl = {k: torch.empty(v, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32) for k, v in keys.items()}
    return synth_state_dict(tmpl, seed=0, mode=mode)


def test_lafc_oracle_matches_golden():
    g = load_golden("lafc_vanilla_64x96.npz")
    flow, edge = LO.lafc_forward(_sd("lafc_vanilla_state_keys.json"), lafc_model.DEFAULT_CONFIG, g["flows"], g["masks"])
    assert max_err(flow, g["flow"]) < 1e-5 and max_err(edge, g["edge"]) < 1e-5


def test_raft_oracle_matches_golden():
    g = load_golden("raft_128x160_it6.npz")
    lo, up = RO.raft_forward(_sd("raft_state_keys.json"), g["image1"], g["image2"], iters=6)
    assert rel_err(lo, g["flow_low"]) < 1e-5 and rel_err(up, g["flow_up"]) < 1e-5


def test_warp_oracle_matches_golden():
    g = load_golden("warp_24x40.npz")
    assert max_err(RO.image_warp(g["img"], g["flow"]), g["warped"]) < 1e-6
    o1, o2 = RO.fb_consistency(g["f1"], g["f2"])
    assert torch.equal(o1, g["occ_fw"]) and torch.equal(o2, g["occ_bw"])


@pytest.fixture()
def fake(monkeypatch):
    for m in (lafc_model, raft_model):
        monkeypatch.setattr(m, "ops", fake_ops)
        monkeypatch.setattr(m, "PackedConv", fake_ops.PackedConv)


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("ct", ["vanilla", "gated"])
def test_lafc_host_logic_and_keys(fake, ct, mode, monkeypatch):
    """mode 'bf16x3' walks the split-chain plumbing (conv -> conv hand-overs as ops.Split, fp32 forms for residual operands and the
    Cout <= 4 convs) over the CPU spec, whose Splits carry exact values: the result must not change."""
    monkeypatch.setattr(fake_ops, "DEFAULT_CONV_PRECISION", mode)
    keys = json.load(open(os.path.join(GOLDEN, f"lafc_{ct}_state_keys.json")))
    m = lafc_model.Model(dict(lafc_model.DEFAULT_CONFIG, conv_type=ct)).eval()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == keys
    m.load_state_dict(_sd(f"lafc_{ct}_state_keys.json"), strict=True)
    g = load_golden(f"lafc_{ct}_64x96.npz")
    flow, edge = m(g["flows"], g["masks"])
    assert rel_err(flow, g["flow"]) < 2e-5 and max_err(edge, g["edge"]) < 2e-5


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_raft_host_logic_and_keys(fake, mode, monkeypatch):
    monkeypatch.setattr(fake_ops, "DEFAULT_CONV_PRECISION", mode)
    keys = json.load(open(os.path.join(GOLDEN, "raft_state_keys.json")))
    m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == keys
    m.load_state_dict(_sd("raft_state_keys.json"), strict=True)
    g = load_golden("raft_128x160_it6.npz")
    lo, up = m(g["image1"], g["image2"], iters=6, test_mode=True)
    assert rel_err(lo, g["flow_low"]) < 1e-4 and rel_err(up, g["flow_up"]) < 1e-4


@pytest.mark.skipif(not RL.available(), reason="reference tree not mounted")
def test_flow_oracles_match_live_reference():
    ref = RL.lafc_model(dict(lafc_model.DEFAULT_CONFIG))
    sd = _sd("lafc_vanilla_state_keys.json")
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(2)
    fl, ms = torch.randn(1, 2, 3, 32, 48, generator=g), (torch.rand(1, 1, 3, 32, 48, generator=g) > 0.5).float()
    a, b = ref(fl, ms), LO.lafc_forward(sd, lafc_model.DEFAULT_CONFIG, fl, ms)
    assert max_err(a[0], b[0]) < 1e-6 and max_err(a[1], b[1]) < 1e-6
    iw, fb = RL.warp_fns()
    img, f = torch.randn(1, 3, 20, 28, generator=g), torch.randn(1, 2, 20, 28, generator=g) * 2
    assert max_err(iw(img, f), RO.image_warp(img, f)) == 0


def test_flow_pipeline_batched_pairs_match_per_pair_calls(fake):
    """fgt_amd.flow_pipeline.compute_flows (per-frame encoder cache + batched pairs) == per-pair RAFT calls in the
    reference's order (tool/video_inpainting.py:246-263), on the CPU kernel spec."""
    from fgt_amd import flow_pipeline
    m = raft_model.RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    m.load_state_dict(_sd("raft_state_keys.json"), strict=True)
    g = load_golden("raft_128x160_it6.npz")
    frames = torch.cat([g["image1"], g["image2"], g["image1"].flip(-1)], 0)
    fw, bw = flow_pipeline.compute_flows(m, frames, iters=3, batch=3, enc_batch=2)
    for i in range(2):
        assert rel_err(fw[i:i + 1], m(frames[i:i + 1], frames[i + 1:i + 2], iters=3, test_mode=True)[1]) < 1e-5
        assert rel_err(bw[i:i + 1], m(frames[i + 1:i + 2], frames[i:i + 1], iters=3, test_mode=True)[1]) < 1e-5


def test_complete_flows_batched_matches_reference_loop(fake):
    """fgt_amd.flow_pipeline.complete_flows == the per-pivot loop of tool/video_inpainting.py:367-384 (run with the oracle)."""
    from fgt_amd import flow_pipeline
    cfg = lafc_model.DEFAULT_CONFIG
    sd = _sd("lafc_vanilla_state_keys.json")
    m = lafc_model.Model(dict(cfg)).eval()
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(4)
    t, H, W = 7, 32, 48
    flows = torch.randn(1, 2, t, H, W, generator=g)
    masks = (torch.rand(1, 1, t, H, W, generator=g) > 0.6).float()
    diffused = flows * (1 - masks)
    got = flow_pipeline.complete_flows(m, flows, masks, diffused, batch=3)
    for i in range(t):
        ind = LO.indices_gen(i, 3, 3, t)
        assert ind == flow_pipeline.indices_gen(i, 3, 3, t)
        o = LO.lafc_forward(sd, cfg, diffused[:, :, ind], masks[:, :, ind])[0]
        ref = o * masks[:, :, ind][:, :, 1] + flows[:, :, ind][:, :, 1] * (1 - masks[:, :, ind][:, :, 1])
        assert rel_err(got[i:i + 1], ref) < 2e-5


def test_complete_flows_runs_the_diffusion_fill_itself(fake, monkeypatch):
    """complete_flows(diffused=None) == diffusion() (tool/video_inpainting.py:42-51, restated in oracle/fill_oracle.py) + the LAFC loop:
    the flow-completion stage from raw flows, as `complete_flow` (:341-386) runs it."""
    import numpy as np
    from fgt_amd import flow_pipeline
    from oracle import fill_oracle as FO
    monkeypatch.setattr(flow_pipeline, "ops", fake_ops)
    cfg = lafc_model.DEFAULT_CONFIG
    sd = _sd("lafc_vanilla_state_keys.json")
    m = lafc_model.Model(dict(cfg)).eval()
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    t, H, W = 5, 32, 48
    flows = torch.randn(1, 2, t, H, W, generator=g)
    masks = torch.zeros(1, 1, t, H, W)
    for i in range(t):
        masks[0, 0, i, 8 + i:20 + i, 10:30] = 1
    # the reference's layout: flows [t,H,W,2], masks [t,H,W,1]
    ref_d = FO.diffusion(flows[0].permute(1, 2, 3, 0).numpy(), masks[0].permute(1, 2, 3, 0).numpy())
    d = flow_pipeline.diffusion(flows, masks)
    assert d.shape == flows.shape
    assert np.abs(d[0].permute(1, 2, 3, 0).numpy() - ref_d).max() < 1e-5
    a = flow_pipeline.complete_flows(m, flows, masks, None, batch=2)
    b = flow_pipeline.complete_flows(m, flows, masks, d, batch=2)
    assert torch.equal(a, b)
