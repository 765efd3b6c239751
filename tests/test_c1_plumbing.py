"""BASELINE config #1 ("tool plumbing"): the reference tool's driver steps over the drop-in boundary.

CPU (here): tests/tool_driver.py — the reduced restatement of tool/video_inpainting.py's initialize_* / calculate_flow /
complete_flow / FGT stage — is pinned to the reference's OWN function bodies (cut out of the tool with ast; the tool itself cannot
be imported: cv2 / torchvision / np.bool) using recording stand-in models: same calls, same arguments, same outputs.
GPU (-m gpu): the driver runs over the fgt_amd drop-ins exactly as the tool would reach them — `sys.path.insert(0, fgt_amd/dropin)`,
`import_module("FGT.models.model").Model(configs)`, `import_module("LAFC.models.lafc")`, `from RAFT import RAFT` wrapped in
DataParallel, checkpoints + yaml on disk, strict load_state_dict — RAFT -> diffusion -> LAFC -> gradient propagation -> Poisson
blend -> FGT on a 20-frame synthetic clip, and every stage is compared with the restructured fast path the benchmark uses
(`flow_pipeline.compute_flows` / `complete_flows`, `propagation` / `blending` clip calls, `scheduler.ClipRunner`).
"""
import argparse
import ast
import json
import os
import sys

import numpy as np
import pytest
import torch

import tool_driver as TD
from oracle import reference_glue as RG
from util import GOLDEN

torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ CPU: driver == reference code
def _ref_functions(names):
    import glob
    import yaml
    from importlib import import_module
    tree = ast.parse(open(RG.TOOL).read(), RG.TOOL)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(d.name for d in defs) == sorted(names)
    ns = {"torch": torch, "np": np, "os": os, "glob": glob, "yaml": yaml, "import_module": import_module, "print": lambda *a, **k: None}
    exec(compile(ast.Module(body=defs, type_ignores=[]), RG.TOOL, "exec"), ns)
    return ns


@pytest.mark.skipif(not RG.available(), reason="reference tree not mounted (GPU box)")
def test_driver_follows_the_reference_tool_code():
    ns = _ref_functions(("calculate_flow", "complete_flow", "diffusion", "np2tensor", "indicesGen", "norm_flows"))
    g = torch.Generator().manual_seed(0)
    N, H, W = 7, 16, 24
    video = torch.rand(N, 3, H, W, generator=g) * 255
    calls = []

    def raft(i1, i2, iters=12, test_mode=False):
        calls.append((float(i1.sum()), float(i2.sum()), iters, test_mode))
        return None, (i1 - i2)[:, :2] * 0.1

    args = argparse.Namespace(imgH=H, imgW=W, vis_flows=False, outroot="")
    for mode in ("forward", "backward"):
        calls.clear()
        want = ns["calculate_flow"](args, raft, video, mode)
        ref_calls = list(calls)
        calls.clear()
        got = TD.calculate_flow(raft, video, mode)
        assert calls == ref_calls and np.array_equal(got, want) and got.shape == (H, W, 2, N - 1)
    # complete_flow: a stand-in regionfill (zero the hole, add 1) and a stand-in LAFC that depends on every input element
    flows = np.random.default_rng(0).normal(size=(H, W, 2, N - 1)).astype(np.float32)
    masks = np.random.default_rng(1).random((H, W, N)) > 0.7
    rf = type("rf", (), {"regionfill": staticmethod(lambda m, k: np.where(k, 1.0, m))})
    ns["rf"] = rf
    lafc = lambda inp, cm: (inp.mean(2) * 0.5 + cm.sum(2) * 0.25, None)
    cfg = {"num_flows": 3, "flow_interval": 3}
    for mode in ("forward", "backward"):
        want = ns["complete_flow"](cfg, lafc, flows, masks, mode, "cpu")
        got = TD.complete_flow(cfg, lafc, flows, masks, mode, "cpu", rf.regionfill)
        assert len(got) == len(want) == N - 1 and all(torch.equal(a, b) for a, b in zip(got, want))
    # the FGT stage against the ast-extracted window loop (frame 5 composed three times)
    n = 23
    fr = torch.rand(1, n, 3, H, W, generator=g)
    ms = (torch.rand(1, n, 1, H, W, generator=g) > 0.5).float()
    vf = np.random.default_rng(2).normal(size=(H, W, 2, n - 1)).astype(np.float32)
    model = lambda mf, fl, m: torch.tanh(mf[0] * 1.1 + fl[0].mean(1, keepdim=True) * 0.4 - m[0] * 0.3)
    got = TD.fgt_stage(model, fr, ms, vf)
    vfn = np.concatenate([np.moveaxis(vf, -1, 0), np.moveaxis(vf, -1, 0)[-1:]], 0)
    flows_t = ns["norm_flows"](ns["np2tensor"](vfn, near="t"))
    want, _ = RG.window_loop()(model, fr, ms, flows_t)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


# ------------------------------------------------------------------------------------------------ GPU: the chain over the drop-ins
def _keys(name):
    keys = json.load(open(os.path.join(GOLDEN, name)))
    return {k: torch.empty(v, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32) for k, v in keys.items()}


@pytest.mark.gpu
def test_tool_chain_over_dropins_matches_fast_paths(dev, tmp_path):
    from fgt_amd import blending, flow_pipeline, propagation
    from fgt_amd.fgt_model import DEFAULT_CONFIG as FGT_CFG
    from fgt_amd.lafc_model import DEFAULT_CONFIG as LAFC_CFG
    from fgt_amd.scheduler import ClipRunner, prepare_flows
    from fgt_amd.synth import synth_clip, synth_state_dict
    sys.path.insert(0, os.path.join(ROOT, "fgt_amd", "dropin"))          # what replaces the reference's sys.path entries (:4-6)
    try:
        fgt_dir, lafc_dir, raft_pth = TD.write_checkpoints(
            str(tmp_path), synth_state_dict(_keys("fgt_vanilla_state_keys.json"), seed=0), dict(FGT_CFG, model="model"),
            synth_state_dict(_keys("lafc_vanilla_state_keys.json"), seed=0, mode="kaiming"), dict(LAFC_CFG, model="lafc", flow_interval=3),
            synth_state_dict(_keys("raft_state_keys.json"), seed=0, mode="kaiming"))
        args = argparse.Namespace(raft_model=raft_pth, lafc_ckpts=lafc_dir, fgt_ckpts=fgt_dir, small=False, mixed_precision=False, alternate_corr=False)
        RAFT_model = TD.initialize_RAFT(args, dev)
        LAFC_model, LAFC_config = TD.initialize_LAFC(args, dev)
        FGT_model, FGT_config = TD.initialize_FGT(args, dev)
    finally:
        sys.path.remove(os.path.join(ROOT, "fgt_amd", "dropin"))
    assert type(RAFT_model).__module__ == "fgt_amd.raft_model" and type(LAFC_model).__module__ == "fgt_amd.lafc_model"
    assert type(FGT_model).__module__ == "fgt_amd.fgt_model"

    N, H, W = 20, 128, 160                                                # (RAFT's 4-level correlation pyramid needs H, W >= 128)
    frames01, _, masks = synth_clip(N, H, W, seed=21, device=dev)
    video = frames01[0] * 255.0                                           # [N,3,H,W] 0..255
    # ---- step 2: RAFT pair by pair (tool) vs per-frame encoder cache + batched pairs
    flow_f = TD.calculate_flow(RAFT_model, video, "forward")
    flow_b = TD.calculate_flow(RAFT_model, video, "backward")
    ff, fb = flow_pipeline.compute_flows(RAFT_model, video, iters=20)
    assert np.array_equal(flow_f, ff.permute(2, 3, 1, 0).cpu().numpy()) and np.array_equal(flow_b, fb.permute(2, 3, 1, 0).cpu().numpy())
    # ---- step 4: diffusion + LAFC, one call per pivot (tool) vs one batched diffusion + 8 pivots per LAFC call
    flow_mask = masks[0, :, 0].permute(1, 2, 0).cpu().numpy() > 0          # [H,W,N]

    def regionfill(m, k):                                                  # rf.regionfill's contract served by the device solver
        out = flow_pipeline.diffusion(torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(dev)[None, None, None],
                                      torch.from_numpy(np.ascontiguousarray(k)).to(dev)[None, None, None].float())
        return out[0, 0, 0].cpu().numpy()

    done_f = TD.complete_flow(LAFC_config, LAFC_model, flow_f, flow_mask, "forward", dev, regionfill)
    fl5 = ff.permute(1, 0, 2, 3)[None]                                     # [1,2,N-1,H,W]
    fast_f = flow_pipeline.complete_flows(LAFC_model, fl5, masks[:, :-1].permute(0, 2, 1, 3, 4), num_flows=3, interval=3)
    d = (torch.cat(done_f, 0) - fast_f).abs().max().item()
    print(f"[parity] C1 complete_flow per-pivot (tool order) vs batched: max diff {d:.2e} px (flows up to {fast_f.abs().max().item():.1f} px)")
    assert d <= 1e-4 * max(1.0, fast_f.abs().max().item())                  # per-map CG on one map vs the same CG batched: same iterates
    done_b = TD.complete_flow(LAFC_config, LAFC_model, flow_b, flow_mask, "backward", dev, regionfill)
    videoFlowF = torch.stack(done_f, -1).squeeze(0).permute(1, 2, 0, 3).cpu().numpy()     # tensor2np (:69-71): [H,W,2,N-1]
    videoFlowB = torch.stack(done_b, -1).squeeze(0).permute(1, 2, 0, 3).cpu().numpy()
    # ---- step 5/6: gradient propagation + Poisson blend through the reference signatures vs the clip-level device calls
    vid = frames01[0].permute(2, 3, 1, 0).cpu().numpy()                     # [H,W,3,N]
    hole = flow_mask
    gx = np.concatenate((np.diff(vid, axis=1), np.zeros((H, 1, 3, N), np.float32)), 1)
    gy = np.concatenate((np.diff(vid, axis=0), np.zeros((1, W, 3, N), np.float32)), 0)
    h4 = np.broadcast_to(hole[:, :, None, :], gx.shape)
    gx[h4] = 0
    gy[h4] = 0
    pargs = argparse.Namespace(Nonlocal=False, consistencyThres=5.0, alpha=0.1)
    gxf, gyf, mgrad = propagation.get_flowNN_gradient(pargs, gx, gy, hole, hole, videoFlowF, videoFlowB, None, None)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.moveaxis(a, -1, 0))).to(dev)
    cx, cy, cfill = propagation.propagate_gradients(t(gx), t(gy), t(hole), t(videoFlowF), t(videoFlowB))
    assert np.array_equal(np.moveaxis(gxf, -1, 0), cx.cpu().numpy()) and np.array_equal(np.moveaxis(mgrad, -1, 0), cfill.cpu().numpy())
    b0, u0 = blending.Poisson_blend_img(vid[..., 3], gxf[:, : W - 1, :, 3], gyf[: H - 1, :, :, 3], hole[..., 3], mgrad[..., 3])
    bc, uc = blending.poisson_blend_clip(t(vid), cx, cy, t(hole), cfill)
    assert np.array_equal(u0, uc[3].cpu().numpy()) and np.array_equal(b0, bc[3].cpu().numpy())
    # ---- step 8: FGT window loop like the tool (host compose) vs ClipRunner (feature cache, window batching, device compose)
    comp = TD.fgt_stage(FGT_model, frames01, masks, videoFlowF)
    flows_dev = prepare_flows(torch.from_numpy(np.ascontiguousarray(np.moveaxis(videoFlowF, -1, 0).transpose(0, 3, 1, 2))).to(dev))
    got = ClipRunner(FGT_model, frames01, flows_dev, masks).run().cpu().numpy()
    want = np.stack([np.asarray(c, np.float32) for c in comp], 0)
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"
