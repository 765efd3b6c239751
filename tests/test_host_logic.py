"""Host-logic check without a GPU: run the nn.Module mirrors over the executable kernel specification
(tests/fake_ops.py) and compare with the reference golden vectors / the oracle.  This validates weight re-layouts,
views, two-source concats, fold geometry and call order — everything except the HIP kernels themselves."""
import pytest
import torch

import fake_ops
from fgt_amd import fgt_model
from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
from fgt_amd.synth import synth_state_dict
from util import load_golden, max_err

torch.set_grad_enabled(False)


@pytest.fixture()
def fake(monkeypatch):
    monkeypatch.setattr(fgt_model, "ops", fake_ops)
    monkeypatch.setattr(fgt_model, "PackedConv", fake_ops.PackedConv)


@pytest.mark.parametrize("split_chain", [False, True], ids=["fp32-tensors", "split-chain"])
@pytest.mark.parametrize("name,conv_type", [("fgt_vanilla_64x96x3.npz", "vanilla"), ("fgt_vanilla_48x80x3.npz", "vanilla"),
                                            ("fgt_gated_48x64x2.npz", "gated")])
def test_fgt_host_logic_matches_reference_golden(fake, monkeypatch, name, conv_type, split_chain):
    """split_chain: the bf16x3 mode's plumbing (conv epilogues hand ops.Split tensors to the next conv) over the lossless
    CPU stand-in: the same golden output must come out, i.e. every Split reaches the consumer its fp32 twin would."""
    if split_chain:
        monkeypatch.setattr(fake_ops, "DEFAULT_CONV_PRECISION", "bf16x3")
    g = load_golden(name)
    m = Model(dict(DEFAULT_CONFIG, conv_type=conv_type)).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
    out = m(g["masked_frames"], g["flows"], g["masks"])
    assert max_err(out, g["out"]) < 5e-6


@pytest.mark.parametrize("name,conv_type", [("fgt_vanilla_64x96x3.npz", "vanilla"), ("fgt_vanilla_48x80x3.npz", "vanilla"),
                                            ("fgt_gated_48x64x2.npz", "gated")])
def test_fgt_f16_mode_model_stays_inside_the_bar(fake, monkeypatch, name, conv_type):
    """The 'f16' arithmetic mode over its CPU model (fake_ops rounds every Split to fp16 and the weights of the GEMMs that consume
    one; exact products): walks the fp16 plumbing of the host code (formats follow the mode, fp32-input GEMMs stay unrounded) and
    bounds what the mode costs against the reference's own output — the GPU tests (tests/test_f16_gpu.py) hold the kernels to the
    same numbers.  Bar 1e-3 (BASELINE.json); plain bf16 operands sit at 9.8e-4."""
    monkeypatch.setattr(fake_ops, "DEFAULT_CONV_PRECISION", "f16")
    g = load_golden(name)
    m = Model(dict(DEFAULT_CONFIG, conv_type=conv_type)).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0), strict=True)
    out = m(g["masked_frames"], g["flows"], g["masks"])
    e = max_err(out, g["out"])
    print(f"[parity] f16 mode (CPU model) {name}: max_abs={e:.3e} ref_max={g['out'].abs().max().item():.3e}")
    assert 1e-6 < e < 2e-4          # really rounded, and 5x inside the bar
