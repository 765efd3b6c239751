"""Golden vectors of the tool's glue, produced by EXECUTING THE REFERENCE'S OWN CODE (authoring container only):

    python tests/golden/make_golden_glue.py

`oracle/reference_glue.py` cuts `indicesGen`, `get_ref_index`, `norm_flows` and the sliding-window compose loop
(tool/video_inpainting.py:90-117, 402-407, 710-740) out of the reference source with `ast` and runs them here.
  glue_schedules.json   window schedules (neighbour ids, reference ids) for several clip lengths / num_ref settings, and
                        indicesGen tables
  glue_compose_23.npz   a 23-frame 16x24 clip (frame 5 is composed three times) pushed through the reference's loop with
                        recorded per-window "model outputs" (values in (-1,1), including exact 0/255 and x.999 edge cases):
                        frames01, masks, the per-window outputs and the loop's comp_frames
  glue_norm_flows.npz   norm_flows input / output (incl. a negative-maximum channel)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_glue as RG  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_grad_enabled(False)
    fns = RG.functions()
    loop = RG.window_loop()
    sched = {}
    for n, stride, step, num_ref in [(1, 5, 10, -1), (6, 5, 10, -1), (11, 5, 10, -1), (23, 5, 10, -1), (46, 5, 10, -1), (80, 5, 10, -1),
                                     (160, 5, 10, -1), (40, 5, 10, 4), (33, 3, 7, 2), (80, 5, 10, 6)]:
        rows = []
        for f in range(0, n, stride):
            nb = [i for i in range(max(0, f - stride), min(n, f + stride + 1))]
            rows.append([nb, fns["get_ref_index"](f, nb, n, step, num_ref)])
        sched[f"{n},{stride},{step},{num_ref}"] = rows
    idx = {f"{p},{i},{fr},{t}": fns["indicesGen"](p, i, fr, t) for t in (3, 5, 12, 80) for p in range(0, t, max(1, t // 6))
           for i, fr in ((3, 3), (1, 3), (2, 5))}
    with open(os.path.join(OUT, "glue_schedules.json"), "w") as f:
        json.dump({"schedules": sched, "indicesGen": idx}, f, sort_keys=True)

    # ---- compose loop
    g = torch.Generator().manual_seed(77)
    n, H, W = 23, 16, 24
    frames01 = torch.rand(1, n, 3, H, W, generator=g)
    frames01[0, 0, :, 0, :4] = torch.tensor([0.0, 1.0, 0.999999, 1.0 / 255.0])       # uint8 truncation edge cases of `valid_frame`
    masks = (torch.rand(1, n, 1, H, W, generator=g) > 0.5).float()
    flows = torch.randn(1, n, 2, H, W, generator=g)
    outs = []

    def model(mf, fl, ms):
        t = mf.shape[1]
        o = torch.tanh(torch.randn(t, 3, H, W, generator=g) * 1.5)
        o[0, 0, 0, :6] = torch.tensor([1.0, -1.0, 0.0, 2.0 / 255.0 - 1.0, 0.99999994, -0.99999994])   # (x+1)/2*255 at 255, 0, 127.5, ~1, ...
        outs.append(o.clone())
        return o

    comp, log = loop(model, frames01, masks, flows)
    assert [len(c.shape) for c in comp] == [3] * n
    arrays = {"frames01": frames01.numpy(), "masks": masks.numpy(), "flows": flows.numpy(),
              "comp": np.stack([np.asarray(c, dtype=np.float32) for c in comp], 0),
              "log": np.array(log, dtype=np.int64)}
    for i, o in enumerate(outs):
        arrays[f"out{i}"] = o.numpy()
    np.savez_compressed(os.path.join(OUT, "glue_compose_23.npz"), **arrays)

    # ---- norm_flows
    fl = torch.randn(1, 5, 2, 12, 20, generator=g) * 7
    fl[0, 2, 1] = -fl[0, 2, 1].abs() - 0.5            # negative signed maximum
    np.savez_compressed(os.path.join(OUT, "glue_norm_flows.npz"), flows=fl.numpy(), normed=fns["norm_flows"](fl).numpy())
    print("windows:", log)


if __name__ == "__main__":
    main()
