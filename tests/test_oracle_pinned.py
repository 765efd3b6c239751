"""Pin the CPU oracle: against the golden vectors produced by the reference itself (always) and against the
live reference modules when /root/reference is present (authoring container)."""
import json
import os

import pytest
import torch

from oracle import fgt_oracle as O
from oracle import reference_loader as RL
from fgt_amd.fgt_model import DEFAULT_CONFIG, Model
from fgt_amd.synth import synth_state_dict
from util import GOLDEN, fgt_inputs, load_golden, max_err

torch.set_grad_enabled(False)


def _sd(conv_type="vanilla", seed=0):
    keys = json.load(open(os.path.join(GOLDEN, f"fgt_{conv_type}_state_keys.json")))
    return synth_state_dict({k: torch.empty(v) for k, v in keys.items()}, seed=seed)


@pytest.mark.parametrize("name", ["fgt_vanilla_64x96x3.npz", "fgt_vanilla_48x80x3.npz"])
def test_oracle_matches_golden(name):
    g = load_golden(name)
    out = O.fgt_forward(_sd(), DEFAULT_CONFIG, g["masked_frames"], g["flows"], g["masks"])
    assert max_err(out, g["out"]) < 2e-6


def test_golden_inputs_are_reproducible_from_seed():
    g = load_golden("fgt_vanilla_64x96x3.npz")
    mf, fl, ms = fgt_inputs(64, 96, 3, 11)
    assert torch.equal(mf, g["masked_frames"]) and torch.equal(fl, g["flows"]) and torch.equal(ms, g["masks"])


@pytest.mark.parametrize("conv_type", ["vanilla", "gated"])
def test_state_dict_keys_match_reference(conv_type):
    keys = json.load(open(os.path.join(GOLDEN, f"fgt_{conv_type}_state_keys.json")))
    m = Model(dict(DEFAULT_CONFIG, conv_type=conv_type))
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == keys
    m.load_state_dict(_sd(conv_type), strict=True)


@pytest.mark.skipif(not RL.available(), reason="reference tree not mounted")
def test_oracle_matches_live_reference_blocks():
    m = RL.fgt_submodules()
    t2t = {'kernel_size': (7, 7), 'stride': (3, 3), 'padding': (3, 3), 'output_size': (60, 108)}
    torch.manual_seed(3)
    S = m.SpatialTransformer([20, 36], 512, 256, 4, 8, 4, 40, 0, 720, t2t).eval()
    T = m.TemporalTransformer([20, 36], 512, 4, 2, 40, 0, 720, t2t).eval()
    x, f = torch.randn(2, 720, 512), torch.randn(2, 720, 256)
    sdS = {"p." + k: v for k, v in S.state_dict().items()}
    sdT = {"p." + k: v for k, v in T.state_dict().items()}
    assert max_err(O.spatial_block(x, f, sdS, "p.", 20, 36, (60, 108)), S(x, f, 2, 0, 0, (0, 0))) < 1e-5
    assert max_err(O.temporal_block(x, sdT, "p.", 2, 20, 36, (60, 108)), T(x, 2, 0, 0, (0, 0))) < 1e-5
    # inference-grid path with padding in both attentions (22 x 35 tokens)
    x, f = torch.randn(2, 22 * 35, 512), torch.randn(2, 22 * 35, 256)
    assert max_err(O.spatial_block(x, f, sdS, "p.", 22, 35, (64, 105)), S(x, f, 2, 22, 35, (64, 105))) < 1e-5
    assert max_err(O.temporal_block(x, sdT, "p.", 2, 22, 35, (64, 105)), T(x, 2, 22, 35, (64, 105))) < 1e-5


@pytest.mark.skipif(not RL.available(), reason="reference tree not mounted")
def test_oracle_matches_live_reference_model():
    ref = RL.fgt_model(dict(DEFAULT_CONFIG))
    sd = _sd()
    ref.load_state_dict(sd, strict=True)
    mf, fl, ms = fgt_inputs(48, 80, 2, 5)
    assert max_err(O.fgt_forward(sd, DEFAULT_CONFIG, mf, fl, ms), ref(mf, fl, ms)) < 2e-6


def test_window_schedule_matches_colab_log():
    # FGT_colab.ipynb cell 8 prints (f, #neighbours, #refs) for the 80-frame demo clip
    sched = O.window_schedule(80)
    got = [(len(nb), len(ref)) for nb, ref in sched]
    assert len(got) == 16 and got[0] == (6, 7) and got[1] == (11, 6) and got[2] == (11, 7) and got[-1] == (10, 7)
    ts = [a + b for a, b in got]
    assert sum(ts) == 275 and sum(t * t for t in ts) == 4749
