"""CPU oracle for RAFT (TEST INFRASTRUCTURE ONLY) — functional fp32 restatement of RAFT/raft.py:87-145 (basic model,
test_mode), RAFT/extractor.py:6-56,118-192, RAFT/corr.py:12-60, RAFT/update.py:6-136, RAFT/utils/utils.py:57-82, and
of the flow-warp operators LAFC/models/utils/fbConsistencyCheck.py:8-47.  Pinned by tests/test_oracle_pinned_flow.py."""
import numpy as np
import torch
import torch.nn.functional as F


def _norm(x, sd, p, kind):
    if kind == "instance":
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride, padding)


def res_block(x, sd, p, kind, stride):
    """extractor.py:6-56."""
    y = F.relu(_norm(_conv(x, sd, p + "conv1.", stride, 1), sd, p + "norm1.", kind))
    y = F.relu(_norm(_conv(y, sd, p + "conv2.", 1, 1), sd, p + "norm2.", kind))
    if stride != 1:
        x = _norm(_conv(x, sd, p + "downsample.0.", stride, 0), sd, p + "downsample.1.", kind)
    return F.relu(x + y)


def encoder(x, sd, p, kind):
    """extractor.py:173-192 (BasicEncoder.forward, eval, no dropout)."""
    x = F.relu(_norm(_conv(x, sd, p + "conv1.", 2, 3), sd, p + "norm1.", kind))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = res_block(x, sd, f"{p}{name}.0.", kind, stride)
        x = res_block(x, sd, f"{p}{name}.1.", kind, 1)
    return _conv(x, sd, p + "conv2.")


def bilinear_sampler(img, coords):
    """RAFT/utils/utils.py:57-71."""
    H, W = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    xg = 2 * xg / (W - 1) - 1
    yg = 2 * yg / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], dim=-1), align_corners=True)


def coords_grid(b, h, w):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(b, 1, 1, 1)


def corr_pyramid(f1, f2, levels=4):
    """corr.py:13-27,52-60."""
    b, d, h, w = f1.shape
    c = torch.matmul(f1.view(b, d, h * w).transpose(1, 2), f2.view(b, d, h * w)) / torch.sqrt(torch.tensor(d).float())
    c = c.reshape(b * h * w, 1, h, w)
    pyr = [c]
    for _ in range(levels - 1):
        c = F.avg_pool2d(c, 2, stride=2)
        pyr.append(c)
    return pyr


def corr_lookup(pyr, coords, r=4):
    """corr.py:29-50 (note the transposed window: delta = stack(meshgrid(dy, dx)) is added to (x, y))."""
    coords = coords.permute(0, 2, 3, 1)
    b, h, w, _ = coords.shape
    out = []
    for i, c in enumerate(pyr):
        d = torch.linspace(-r, r, 2 * r + 1)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)
        cl = coords.reshape(b * h * w, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        out.append(bilinear_sampler(c, cl).view(b, h, w, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def update_block(net, inp, corr, flow, sd, p="update_block."):
    """update.py:79-136 (BasicMotionEncoder, SepConvGRU, FlowHead, mask head)."""
    e = p + "encoder."
    cor = F.relu(_conv(corr, sd, e + "convc1."))
    cor = F.relu(_conv(cor, sd, e + "convc2.", 1, 1))
    flo = F.relu(_conv(flow, sd, e + "convf1.", 1, 3))
    flo = F.relu(_conv(flo, sd, e + "convf2.", 1, 1))
    out = F.relu(_conv(torch.cat([cor, flo], 1), sd, e + "conv.", 1, 1))
    x = torch.cat([inp, out, flow], 1)
    g = p + "gru."
    for s, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([net, x], 1)
        z = torch.sigmoid(_conv(hx, sd, g + f"convz{s}.", 1, pad))
        r = torch.sigmoid(_conv(hx, sd, g + f"convr{s}.", 1, pad))
        q = torch.tanh(_conv(torch.cat([r * net, x], 1), sd, g + f"convq{s}.", 1, pad))
        net = (1 - z) * net + z * q
    delta = _conv(F.relu(_conv(net, sd, p + "flow_head.conv1.", 1, 1)), sd, p + "flow_head.conv2.", 1, 1)
    mask = 0.25 * _conv(F.relu(_conv(net, sd, p + "mask.0.", 1, 1)), sd, p + "mask.2.")
    return net, mask, delta


def upsample_flow(flow, mask):
    """raft.py:73-84."""
    n, _, h, w = flow.shape
    mask = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * h, 8 * w)


def raft_forward(sd, image1, image2, iters=12, flow_init=None):
    """raft.py:87-145 with test_mode=True -> (flow_low, flow_up)."""
    image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
    image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
    f = encoder(torch.cat([image1, image2], 0), sd, "fnet.", "instance")
    b = image1.shape[0]
    pyr = corr_pyramid(f[:b].float(), f[b:].float())
    c = encoder(image1, sd, "cnet.", "batch")
    net, inp = torch.tanh(c[:, :128]), torch.relu(c[:, 128:])
    h, w = image1.shape[2] // 8, image1.shape[3] // 8
    coords0, coords1 = coords_grid(b, h, w), coords_grid(b, h, w)
    if flow_init is not None:
        coords1 = coords1 + flow_init
    up = None
    for _ in range(iters):
        corr = corr_lookup(pyr, coords1)
        net, mask, delta = update_block(net, inp, corr, coords1 - coords0, sd)
        coords1 = coords1 + delta
        up = upsample_flow(coords1 - coords0, mask)
    return coords1 - coords0, up


# ------------------------------------------------------------------------------------ flow warp operators
def image_warp(image, flow):
    """LAFC/models/utils/fbConsistencyCheck.py:8-26 (grid_sample default align_corners=False on a linspace(-1,1) grid)."""
    b, c, h, w = image.size()
    fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], dim=1).permute(0, 2, 3, 1)
    X, Y = np.meshgrid(np.linspace(-1, 1, w), np.linspace(-1, 1, h))
    grid = torch.cat((torch.from_numpy(X.astype("float32"))[None, ..., None], torch.from_numpy(Y.astype("float32"))[None, ..., None]), 3)
    return F.grid_sample(image, grid + fl, mode="bilinear", padding_mode="zeros", align_corners=False)


def fb_consistency(flow_fw, flow_bw, alpha1=0.01, alpha2=0.5):
    """fbConsistencyCheck.py:29-47."""
    sq = lambda x: torch.sum(torch.square(x), dim=1, keepdim=True)
    bw_w, fw_w = image_warp(flow_bw, flow_fw), image_warp(flow_fw, flow_bw)
    occ_fw = (sq(flow_fw + bw_w) > alpha1 * (sq(flow_fw) + sq(bw_w)) + alpha2).float()
    occ_bw = (sq(flow_bw + fw_w) > alpha1 * (sq(flow_bw) + sq(fw_w)) + alpha2).float()
    return occ_fw, occ_bw
