"""CPU oracle for LAFC flow completion (TEST INFRASTRUCTURE ONLY) — functional fp32 restatement of
LAFC/models/lafc.py:18-148 (P3DNet, P3DBlock, EdgeDetection) on torch CPU.  Pinned by tests/test_oracle_pinned_flow.py
against golden vectors produced by the reference and against the live reference when present."""
import torch
import torch.nn.functional as F


def c3(x, sd, p, stride=1, padding=0, slope=0.2):
    """Conv3d + LeakyReLU(0.2) (LAFC/models/utils/network_blocks.py:7-43, norm=None)."""
    y = F.conv3d(x, sd[p + "featureConv.weight"], sd.get(p + "featureConv.bias"), stride, padding)
    return y if slope is None else F.leaky_relu(y, slope)


def c2(x, sd, p, stride=1, padding=1, dilation=1, slope=0.2):
    y = F.conv2d(x, sd[p + "featureConv.weight"], sd.get(p + "featureConv.bias"), stride, padding, dilation)
    return y if slope is None else F.leaky_relu(y, slope)


def p3d(x, sd, p, stride, padding, residual):
    """lafc.py:108-125: (1,k,k) conv then (3,1,1) conv, optional identity."""
    y = c3(x, sd, p + "conv1.", (1, stride, stride), (0, padding, padding))
    y = c3(y, sd, p + "conv2.", 1, (1, 0, 0))
    return x + y if residual else y


def lafc_forward(sd, cfg, flows, masks, edges=None):
    """lafc.py:84-105.  flows [b,2,T,H,W], masks [b,1,T,H,W] -> (flow [b,2,H,W], edge [b,1,H,W])."""
    x = torch.cat((flows, masks), 1) if cfg.get("PASSMASK", 1) else flows
    if edges is not None:
        x = torch.cat((x, edges), 1)
    n = "net."
    e2 = p3d(F.pad(x, (2, 2, 2, 2, 0, 0), mode="replicate"), sd, n + "encoder2.1.", 1, 0, 0)
    e2 = p3d(e2, sd, n + "encoder2.2.", 2, 1, 0)
    c_e2pre = c3(e2, sd, n + "condense2.").squeeze(2)
    e4 = p3d(e2, sd, n + "encoder4.0.", 1, 1, cfg.get("use_residual", 1))
    e4 = p3d(e4, sd, n + "encoder4.1.", 2, 1, 0)
    c_e4pre = c3(e4, sd, n + "condense4_pre.").squeeze(2)
    for i in range(cfg.get("resBlocks", 1)):
        e4 = p3d(e4, sd, n + f"res_blocks.{i}.", 1, 1, 1)
    y = c3(e4, sd, n + "condense4_post.").squeeze(2)
    for i, d in enumerate((8, 4, 2, 1)):
        y = c2(y, sd, n + f"middle.{i}.", 1, d, d)
    y = torch.cat((y, c_e4pre), 1)
    y = c2(F.interpolate(y, scale_factor=2), sd, n + "decoder2.0.conv.")
    y = c2(y, sd, n + "decoder2.1.")
    y = c2(y, sd, n + "decoder2.2.")
    y = torch.cat((y, c_e2pre), 1)
    y = c2(F.interpolate(y, scale_factor=2), sd, n + "decoder.0.conv.")
    y = c2(y, sd, n + "decoder.1.")
    flow = c2(y, sd, n + "decoder.2.", slope=None)
    e = n + "edgeDetector."
    pr = c2(flow, sd, e + "projection.")
    ed = c2(pr, sd, e + "mid_layer_1.")
    ed = c2(ed, sd, e + "mid_layer_2.", slope=None)
    ed = F.leaky_relu(pr + ed, 0.01)                                   # nn.LeakyReLU() default slope (lafc.py:137)
    edge = torch.sigmoid(c2(ed, sd, e + "out_layer.", padding=0, slope=None))
    return flow, edge


def indices_gen(pivot, interval, frames, t):
    """tool/video_inpainting.py:90-100 (reflect indexing around the clip ends)."""
    out = []
    for i in range(-(frames // 2), frames // 2 + 1):
        idx = pivot + interval * i
        if idx < 0:
            idx = abs(idx)
        if idx > t - 1:
            idx = 2 * (t - 1) - idx
        out.append(idx)
    return out
