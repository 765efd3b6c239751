"""Reduced restatement of the reference tool's driver (TEST INFRASTRUCTURE ONLY): the steps of tool/video_inpainting.py that touch
the drop-in boundary, written against the reference's import contract so that the same code drives the reference modules (CPU,
authoring container) and the fgt_amd drop-ins (`sys.path.insert(0, "fgt_amd/dropin")`, MI355X).

  initialize_RAFT / initialize_LAFC / initialize_FGT    :186-230   checkpoint + yaml on disk -> module via the reference's import paths
  calculate_flow                                        :233-288   RAFT pair by pair, iters = 20, test_mode (no resize branch: the clip is
                                                                   processed at flow resolution; cv2.resize is outside the hot path)
  complete_flow                                         :342-386   diffusion fill + one LAFC call per pivot flow
  fgt_stage                                             :687-740   norm_flows, window loop, uint8 compose, ordered blend
`tests/test_c1_plumbing.py` pins this file to the reference's own function bodies (cut out with ast, oracle/reference_glue.py) on
CPU with recording stand-in models, then runs it over the drop-ins on the GPU and compares with the restructured fast paths
(`flow_pipeline.compute_flows` / `complete_flows`, `scheduler.ClipRunner`).
"""
import glob
import os
from importlib import import_module

import numpy as np
import torch
import yaml


def initialize_RAFT(args, device):
    from RAFT import RAFT
    model = torch.nn.DataParallel(RAFT(args))
    model.load_state_dict(torch.load(args.raft_model, map_location=torch.device(device)))
    model = model.module
    model.to(device)
    model.eval()
    return model


def _from_ckpt_dir(pkg, ckpt_dir, device):
    assert len(os.listdir(ckpt_dir)) == 2
    checkpoint, config_file = glob.glob(os.path.join(ckpt_dir, "*.tar"))[0], glob.glob(os.path.join(ckpt_dir, "*.yaml"))[0]
    with open(config_file, "r") as f:
        configs = yaml.full_load(f)
    net = import_module("{}.models.{}".format(pkg, configs["model"]))
    model = net.Model(configs).to(device)
    state = torch.load(checkpoint, map_location=torch.device(device))
    model.load_state_dict(state["model_state_dict"])
    return model, configs


def initialize_LAFC(args, device):
    return _from_ckpt_dir("LAFC", args.lafc_ckpts, device)


def initialize_FGT(args, device):
    return _from_ckpt_dir("FGT", args.fgt_ckpts, device)


def calculate_flow(model, video, mode):
    """video [N,3,H,W] 0..255 -> Flow [H,W,2,N-1] numpy (tool/video_inpainting.py:233-288)."""
    flows = []
    with torch.no_grad():
        for i in range(video.shape[0] - 1):
            if mode == "forward":
                image1, image2 = video[i, None], video[i + 1, None]
            else:
                image1, image2 = video[i + 1, None], video[i, None]
            _, flow = model(image1, image2, iters=20, test_mode=True)
            flows.append(flow[0].permute(1, 2, 0).cpu().numpy())
    return np.stack(flows, -1)


def indicesGen(pivot, interval, frames, t):
    out = []
    for i in range(-(frames // 2), frames // 2 + 1):
        index = pivot + interval * i
        if index < 0:
            index = abs(index)
        if index > t - 1:
            index = 2 * (t - 1) - index
        out.append(index)
    return out


def complete_flow(config, flow_model, flows, flow_masks, mode, device, regionfill):
    """flows [H,W,2,N-1], flow_masks [H,W,N] -> list of [1,2,H,W] completed flows (tool/video_inpainting.py:342-386);
    `regionfill(map, mask)` is rf.regionfill (tool/utils/region_fill.py) or a stand-in with the same contract."""
    flow_masks = np.moveaxis(flow_masks, -1, 0)
    flows = np.moveaxis(flows, -1, 0)
    if flow_masks.ndim == 3:
        flow_masks = flow_masks[:, :, :, np.newaxis]
    flow_masks = flow_masks[0:-1] if mode == "forward" else flow_masks[1:]
    num_flows, flow_interval = config["num_flows"], config["flow_interval"]
    diffused = []
    for i in range(flows.shape[0]):
        f = np.zeros(flows[i].shape)
        f[:, :, 0] = regionfill(flows[i][:, :, 0], flow_masks[i][:, :, 0])
        f[:, :, 1] = regionfill(flows[i][:, :, 1], flow_masks[i][:, :, 0])
        diffused.append(f)
    to_t = lambda a: torch.from_numpy(np.transpose(np.stack(a, 0) if isinstance(a, list) else a, (3, 0, 1, 2))).unsqueeze(0).float().to(device)
    flows_t, masks_t, diff_t = to_t(flows), to_t(flow_masks), to_t(diffused)
    t = diff_t.shape[2]
    filled = [None] * t
    pivot = num_flows // 2
    for i in range(t):
        idx = indicesGen(i, flow_interval, num_flows, t)
        cand_masks = masks_t[:, :, idx]
        with torch.no_grad():
            out = flow_model(diff_t[:, :, idx], cand_masks)
        if isinstance(out, (tuple, list)):
            out = out[0]
        pm = cand_masks[:, :, pivot]
        filled[i] = out * pm + flows_t[:, :, idx][:, :, pivot] * (1 - pm)
    return filled


def fgt_stage(FGT_model, frames_first, masks, videoFlowF, neighbor_stride=5, ref_length=10, num_ref=-1):
    """frames_first [1,N,3,H,W] in [0,1] (device), masks [1,N,1,H,W], videoFlowF [H,W,2,N-1] numpy -> comp frames list
    (tool/video_inpainting.py:697-740)."""
    device = frames_first.device
    n = frames_first.shape[1]
    normed = frames_first * 2 - 1
    comp = [None] * n
    vf = np.moveaxis(videoFlowF, -1, 0)
    vf = np.concatenate([vf, vf[-1:, ...]], axis=0)                                     # :705
    flows = torch.from_numpy(np.transpose(vf, (0, 3, 1, 2))).unsqueeze(0).float()
    fm = torch.max(flows.flatten(3), dim=-1, keepdim=True)[0]                           # norm_flows :402-407
    flows = (flows / fm.unsqueeze(-1)).to(device)
    for f in range(0, n, neighbor_stride):
        nb = [i for i in range(max(0, f - neighbor_stride), min(n, f + neighbor_stride + 1))]
        if num_ref == -1:
            ref = [i for i in range(0, n, ref_length) if i not in nb]
        else:
            ref = []
            for i in range(max(0, f - ref_length * (num_ref // 2)), min(n, f + ref_length * (num_ref // 2)) + 1, ref_length):
                if i not in nb:
                    if len(ref) > num_ref:
                        break
                    ref.append(i)
        sel_m = masks[:, nb + ref]
        with torch.no_grad():
            filled = FGT_model(normed[:, nb + ref] * (1 - sel_m), flows[:, nb + ref], sel_m)
        filled = ((filled + 1) / 2).cpu().permute(0, 2, 3, 1).numpy() * 255
        for i, idx in enumerate(nb):
            valid = frames_first[0, idx].cpu().permute(1, 2, 0).numpy() * 255.0
            m = masks[0, idx].cpu().permute(1, 2, 0).numpy()
            c = np.array(filled[i]).astype(np.uint8) * m + np.array(valid).astype(np.uint8) * (1 - m)
            comp[idx] = c if comp[idx] is None else comp[idx].astype(np.float32) * 0.5 + c.astype(np.float32) * 0.5
    return comp


def write_checkpoints(root, fgt_sd, fgt_cfg, lafc_sd, lafc_cfg, raft_sd):
    """Lay down what the tool's initialize_* expect: <root>/fgt/{*.tar,*.yaml}, <root>/lafc/{*.tar,*.yaml}, <root>/raft.pth
    (DataParallel key names)."""
    for name, sd, cfg in (("fgt", fgt_sd, fgt_cfg), ("lafc", lafc_sd, lafc_cfg)):
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        torch.save({"model_state_dict": sd}, os.path.join(d, "ckpt.pth.tar"))
        with open(os.path.join(d, "config.yaml"), "w") as f:
            yaml.safe_dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, f)
    torch.save({"module." + k: v for k, v in raft_sd.items()}, os.path.join(root, "raft.pth"))
    return os.path.join(root, "fgt"), os.path.join(root, "lafc"), os.path.join(root, "raft.pth")
