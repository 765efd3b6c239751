"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF (authoring container only).

    python tests/golden/make_golden.py

Every fixture stores the seeded inputs and the reference module's fp32 CPU outputs.  Weights are not
stored: they are `fgt_amd.synth.synth_state_dict(<reference state_dict>, seed)` — a pure function of the
key names and shapes, reproducible on the GPU box.  State-dict key/shape lists are stored as JSON so the
drop-in contract (strict load_state_dict) is checked without the reference.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_loader as RL  # noqa: E402
from fgt_amd.synth import synth_state_dict  # noqa: E402
from fgt_amd.fgt_model import DEFAULT_CONFIG  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def fgt_inputs(H, W, t, seed):
    g = torch.Generator().manual_seed(seed)
    fr = torch.rand(1, t, 3, H, W, generator=g) * 2 - 1
    ms = (torch.rand(1, t, 1, H // 8, W // 8, generator=g) > 0.7).float()
    ms = torch.nn.functional.interpolate(ms.view(t, 1, H // 8, W // 8), size=(H, W), mode="nearest").view(1, t, 1, H, W)
    fl = torch.randn(1, t, 2, H, W, generator=g)
    return fr * (1 - ms), fl, ms


def keys_json(model, name):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump({k: list(v.shape) for k, v in model.state_dict().items()}, f, indent=0, sort_keys=True)


def make_fgt():
    torch.set_grad_enabled(False)
    for conv_type in ("vanilla", "gated"):
        cfg = dict(DEFAULT_CONFIG, conv_type=conv_type)
        ref = RL.fgt_model(cfg)
        keys_json(ref, f"fgt_{conv_type}_state_keys.json")
        sd = synth_state_dict(ref.state_dict(), seed=0)
        ref.load_state_dict(sd, strict=True)
        cases = [(64, 96, 3, 11), (48, 80, 3, 12)] if conv_type == "vanilla" else [(48, 64, 2, 13)]
        for H, W, t, seed in cases:
            mf, fl, ms = fgt_inputs(H, W, t, seed)
            out = ref(mf, fl, ms)
            np.savez_compressed(os.path.join(OUT, f"fgt_{conv_type}_{H}x{W}x{t}.npz"), masked_frames=mf.numpy(),
                                flows=fl.numpy(), masks=ms.numpy(), out=out.numpy(), seed=np.int64(0))
            print("fgt", conv_type, H, W, t, float(out.abs().max()))
    # one trained-grid (240x432) case, t = 2, reference-style N(0, 0.02) weights (the BASELINE parity config)
    cfg = dict(DEFAULT_CONFIG)
    ref = RL.fgt_model(cfg)
    ref.load_state_dict(synth_state_dict(ref.state_dict(), seed=0), strict=True)
    mf, fl, ms = fgt_inputs(240, 432, 2, 14)
    out = ref(mf, fl, ms)
    # inputs are regenerated from the seed by the test (same fgt_inputs recipe); only the output is stored
    np.savez_compressed(os.path.join(OUT, "fgt_vanilla_240x432x2.npz"), out=out.numpy(), seed=np.int64(0))
    print("fgt trained grid", float(out.abs().max()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["fgt"]
    if "fgt" in which:
        make_fgt()
    if "lafc" in which or "raft" in which or "warp" in which:
        from make_golden_flow import make_flow  # noqa
        make_flow(which)
