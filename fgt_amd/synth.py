"""Deterministic synthetic weights and inputs (no datasets / checkpoints are reachable).

`synth_state_dict` fills a state_dict *by key name* from per-key seeded CPU generators, so the reference
model (in the authoring container) and the MI355X model (on the GPU box) get bit-identical weights
without shipping them: the golden fixtures under tests/golden/ were produced with these weights.
"""
import zlib

import torch


def synth_state_dict(template, seed=0, mode="normal"):
    """template: mapping key -> tensor (shapes/dtypes).  Returns a new dict of CPU fp32 tensors.

    mode 'normal'  : weights ~ N(0, 0.02) like the reference's FGT init (FGT/models/BaseNetwork.py:20-46),
                     biases ~ N(0, 0.02) (non-zero so bias paths are exercised), norm weights 1 + N(0, 0.02).
    mode 'kaiming' : conv/linear weights ~ N(0, sqrt(2/fan_in)) (LAFC/models/BaseNetwork.py:25-51 flavour) so
                     activations stay O(1) through deep stacks.
    """
    out = {}
    for key in sorted(template.keys()):
        ref = template[key]
        if not torch.is_floating_point(ref):
            out[key] = torch.zeros_like(ref, device="cpu")
            continue
        alias = key.replace("downsample.1.", "norm3.")   # RAFT/extractor.py:41-43: one module under two names
        g = torch.Generator().manual_seed((zlib.crc32(alias.encode()) + 7919 * seed) % (2 ** 31))
        r = torch.randn(tuple(ref.shape), generator=g, dtype=torch.float32)
        is_norm = any(s in alias for s in ("norm", "bn")) and ref.dim() == 1
        if key.endswith("running_var"):
            t = 1.0 + 0.2 * r.abs()
        elif key.endswith("running_mean"):
            t = 0.1 * r
        elif is_norm and key.endswith("weight"):
            t = 1.0 + 0.02 * r
        elif ref.dim() >= 2 and mode == "kaiming":
            fan_in = ref[0].numel()
            t = r * (2.0 / fan_in) ** 0.5
        else:
            t = 0.02 * r
        out[key] = t
    return out


def synth_clip(n_frames, H=240, W=432, seed=1234, device="cpu"):
    """SURVEY.md §8d synthetic clip: frames01 [1,N,3,H,W] in [0,1], binary masks [1,N,1,H,W] (a box of ~16 % area
    drifting 2 px/frame), flows [1,N,2,H,W] smooth noise already divided by the per-(frame, channel) signed max
    (tool/video_inpainting.py:402-407)."""
    g = torch.Generator().manual_seed(seed)
    frames = torch.rand(1, n_frames, 3, H, W, generator=g)
    # low-pass so that frames look like images rather than white noise
    frames = torch.nn.functional.avg_pool2d(frames.view(-1, 1, H, W), 5, 1, 2).view(1, n_frames, 3, H, W)
    frames = (frames - frames.amin()) / (frames.amax() - frames.amin())
    masks = torch.zeros(1, n_frames, 1, H, W)
    bh, bw = int(H * 0.4), int(W * 0.4166)
    for i in range(n_frames):
        y0 = (H - bh) // 2 + int(8 * torch.sin(torch.tensor(i / 7.0)))
        x0 = (W // 6 + 2 * i) % max(1, W - bw)
        masks[0, i, 0, y0:y0 + bh, x0:x0 + bw] = 1.0
    flows = torch.randn(1, n_frames, 2, H // 8, W // 8, generator=g)
    flows = torch.nn.functional.interpolate(flows.view(-1, 2, H // 8, W // 8), size=(H, W), mode="bilinear",
                                            align_corners=False).view(1, n_frames, 2, H, W)
    fmax = flows.flatten(3).max(dim=-1, keepdim=True)[0]
    flows = flows / fmax.unsqueeze(-1)
    return frames.to(device), flows.to(device), masks.to(device)
