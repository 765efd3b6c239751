"""MI355X-native RAFT optical-flow estimator behind the reference's nn.Module API.

`RAFT(args).forward(image1[B,3,H,W] 0..255, image2, iters=12, flow_init=None, upsample=True, test_mode=False)`
with the constructor/forward signature and state_dict keys of RAFT/raft.py:24-145 (basic model, `args.small=False`;
`tool/video_inpainting.py:186-197,263`).  Returns `(flow_low, flow_up)` in test_mode, the list of up-sampled
predictions otherwise.

Feature / context encoders, the all-pairs correlation GEMM, the motion encoder, the separable ConvGRU (sigmoid/tanh and
the gate blend fused into conv epilogues, `cat([h, x])` read as two sources), flow/mask heads: fgt_conv2d.  Instance
norm, correlation pyramid pooling, the 4-level 9x9 bilinear lookup and convex up-sampling: dedicated HBM-bound kernels.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .fgt_model import ConvParams
from .ops import PackedConv


class BatchNormParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps = 1e-5


class NoParams(nn.Module):
    """InstanceNorm2d(affine=False) / activation placeholders: no state."""


def _conv(cin, cout, k):
    m = ConvParams(cin, cout, k)
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return m


class ResBlockParams(nn.Module):
    """RAFT/extractor.py:6-44 attribute names (norm3 is shared with downsample.1)."""

    def __init__(self, cin, cout, norm, stride):
        super().__init__()
        self.conv1, self.conv2 = _conv(cin, cout, 3), _conv(cout, cout, 3)
        mk = (lambda: BatchNormParams(cout)) if norm == "batch" else NoParams
        self.norm1, self.norm2 = mk(), mk()
        self.stride = stride
        if stride != 1:
            self.norm3 = mk()
            self.downsample = nn.ModuleList([_conv(cin, cout, 1), self.norm3])
        else:
            self.downsample = None


class EncoderParams(nn.Module):
    """RAFT/extractor.py:118-171 (BasicEncoder)."""

    def __init__(self, output_dim, norm):
        super().__init__()
        self.norm_fn = norm
        self.norm1 = BatchNormParams(64) if norm == "batch" else NoParams()
        self.conv1 = _conv(3, 64, 7)
        self.layer1 = nn.ModuleList([ResBlockParams(64, 64, norm, 1), ResBlockParams(64, 64, norm, 1)])
        self.layer2 = nn.ModuleList([ResBlockParams(64, 96, norm, 2), ResBlockParams(96, 96, norm, 1)])
        self.layer3 = nn.ModuleList([ResBlockParams(96, 128, norm, 2), ResBlockParams(128, 128, norm, 1)])
        self.conv2 = _conv(128, output_dim, 1)


class MotionEncoderParams(nn.Module):
    def __init__(self, cor_planes):
        super().__init__()
        self.convc1, self.convc2 = ConvParams(cor_planes, 256, 1), ConvParams(256, 192, 3)
        self.convf1, self.convf2 = ConvParams(2, 128, 7), ConvParams(128, 64, 3)
        self.conv = ConvParams(64 + 192, 128 - 2, 3)


class SepConvGRUParams(nn.Module):
    def __init__(self, hidden, inp):
        super().__init__()
        for n in ("z", "r", "q"):
            setattr(self, f"conv{n}1", ConvParams(hidden + inp, hidden, (1, 5)))
        for n in ("z", "r", "q"):
            setattr(self, f"conv{n}2", ConvParams(hidden + inp, hidden, (5, 1)))


class FlowHeadParams(nn.Module):
    def __init__(self, cin, hidden):
        super().__init__()
        self.conv1, self.conv2 = ConvParams(cin, hidden, 3), ConvParams(hidden, 2, 3)


class UpdateBlockParams(nn.Module):
    def __init__(self, cor_planes, hidden=128):
        super().__init__()
        self.encoder = MotionEncoderParams(cor_planes)
        self.gru = SepConvGRUParams(hidden, 128 + hidden)
        self.flow_head = FlowHeadParams(hidden, 256)
        self.mask = nn.ModuleList([ConvParams(128, 256, 3), NoParams(), ConvParams(256, 64 * 9, 1)])


class RAFT(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        if getattr(args, "small", False):
            raise NotImplementedError("RAFT-small is not used by FGT (tool/video_inpainting.py:186-197 loads raft-things)")
        if getattr(args, "alternate_corr", False):
            raise NotImplementedError("alternate_corr is dead code in the reference (RAFT/raft.py:106 names an undefined class)")
        self.hidden_dim = self.context_dim = 128
        self.corr_levels, self.corr_radius = 4, 4
        args.corr_levels, args.corr_radius = 4, 4                      # raft.py:36-37
        if not hasattr(args, "dropout"):
            args.dropout = 0
        self.fnet = EncoderParams(256, "instance")
        self.cnet = EncoderParams(256, "batch")
        self.update_block = UpdateBlockParams(self.corr_levels * (2 * self.corr_radius + 1) ** 2)
        self._packed, self._key = None, None
        self.use_graph = os.environ.get("FGT_GRAPHS", "0") == "1"
        self.hoist_context = os.environ.get("FGT_RAFT_HOIST", "1") != "0"
        self.fuse_zr = os.environ.get("FGT_RAFT_ZR", "1") != "0"                 # z || r of a GRU half as one two-headed conv (needs the hoisted context)
        self.batched_corr = os.environ.get("FGT_RAFT_BCORR", "1") != "0"         # the pair batch's correlation volumes as ONE batched GEMM launch
        self._graphs = {}

    # ------------------------------------------------------------------ packing
    @staticmethod
    def _pk(conv, bn=None):
        """PackedConv with an eval-mode BatchNorm folded into the epilogue's per-channel scale/bias."""
        w, b = conv.weight.detach(), conv.bias.detach()
        if bn is None or isinstance(bn, NoParams):
            return PackedConv(w, b)
        s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        return PackedConv(w, (b - bn.running_mean) * s + bn.bias.detach(), scale=s)

    def _pack_encoder(self, e):
        P = {"conv1": self._pk(e.conv1, e.norm1), "conv2": self._pk(e.conv2), "blocks": []}
        for layer in (e.layer1, e.layer2, e.layer3):
            for rb in layer:
                P["blocks"].append(dict(c1=self._pk(rb.conv1, rb.norm1), c2=self._pk(rb.conv2, rb.norm2), stride=rb.stride,
                                        down=None if rb.downsample is None else self._pk(rb.downsample[0], rb.norm3)))
        return P

    def packed(self):
        key = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._packed is None or key != self._key:
            u = self.update_block
            P = {"fnet": self._pack_encoder(self.fnet), "cnet": self._pack_encoder(self.cnet)}
            P["enc"] = {n: self._pk(getattr(u.encoder, n)) for n in ("convc1", "convc2", "convf1", "convf2", "conv")}
            # the same conv with 2 zero output channels appended (126 -> 128): in the split chain its epilogue writes the GRU input buffer's
            # channels 128..255 as one aligned 128-channel slice (relu(0) = 0 lands in the two flow channels, which are written right after)
            cw, cb = u.encoder.conv.weight.detach(), u.encoder.conv.bias.detach()
            P["enc"]["conv128"] = PackedConv(torch.cat([cw, torch.zeros(2, *cw.shape[1:], device=cw.device, dtype=cw.dtype)], 0),
                                             torch.cat([cb, torch.zeros(2, device=cb.device, dtype=cb.dtype)], 0))
            # convc1 with its input channels zero-padded 324 -> 352 (a multiple of 32): in the split chain the correlation lookup writes its
            # taps as a split tensor (zeros in the padding) and this 1x1 conv runs on the LDS-DMA kernel instead of the register-staged one
            c1w, c1b = u.encoder.convc1.weight.detach(), u.encoder.convc1.bias.detach()
            cpad = (-c1w.shape[1]) % 32
            P["enc"]["convc1p"] = PackedConv(torch.cat([c1w, torch.zeros(c1w.shape[0], cpad, 1, 1, device=c1w.device, dtype=c1w.dtype)], 1), c1b)
            # SepConvGRU (update.py:36-60): hx = [h | inp | motion]; `inp` = relu(cnet[:, 128:]) does not change over the refinement loop
            # (raft.py:112-115, 126-128), so each conv is split into the per-iteration part over [h | motion] (256 of the 384 input channels)
            # and the context part over `inp` (+ bias), which iterate() evaluates ONCE per pair and hands to the per-iteration conv as a
            # bias map (fgt_conv_desc.ld_bias): a third of the GRU's multiply-adds leaves the loop.  FGT_RAFT_HOIST=0: the undivided convs.
            P["gru"], P["gru_it"], P["gru_ctx"] = {}, {}, {}
            for n in ("convz1", "convr1", "convq1", "convz2", "convr2", "convq2"):
                cv = getattr(u.gru, n)
                w, b = cv.weight.detach(), cv.bias.detach()
                hd = self.hidden_dim
                P["gru"][n] = self._pk(cv)
                P["gru_it"][n] = PackedConv(torch.cat([w[:, :hd], w[:, 2 * hd:]], 1), None)
                P["gru_ctx"][n] = PackedConv(w[:, hd:2 * hd], b)
            # z and r of a GRU half read the same hx (update.py:46-49, 53-56): ONE conv with the output channels [z | r] and a two-headed epilogue
            # (fgt_conv_desc.dual_n0: sigmoid -> z as fp32, sigmoid * h -> r * h as a split tensor) instead of two Cout = 128 launches — the im2col
            # rows are fetched once and the tiles are 256 channels wide.  FGT_RAFT_ZR=0: the two launches.
            for s_ in ("1", "2"):
                wz, wr = getattr(u.gru, "convz" + s_).weight.detach(), getattr(u.gru, "convr" + s_).weight.detach()
                bz, br = getattr(u.gru, "convz" + s_).bias.detach(), getattr(u.gru, "convr" + s_).bias.detach()
                wzr = torch.cat([wz, wr], 0)
                P["gru_it"]["convzr" + s_] = PackedConv(torch.cat([wzr[:, :hd], wzr[:, 2 * hd:]], 1), None)
                P["gru_ctx"]["convzr" + s_] = PackedConv(wzr[:, hd:2 * hd], torch.cat([bz, br], 0))
            P["fh"] = (self._pk(u.flow_head.conv1), self._pk(u.flow_head.conv2))
            P["mask"] = (self._pk(u.mask[0]), self._pk(u.mask[2]))
            self._packed, self._key = P, key
        return self._packed

    # ------------------------------------------------------------------ encoders
    def _encode(self, x, P, instance):
        """BasicEncoder.forward (extractor.py:173-192) on a channels-last batch.
        bf16x3 (round 6): the encoders run as SPLIT CHAINS like the update block — an activation that feeds a convolution is handed over as a bf16 hi / lo
        pair written by its producer (the conv epilogue, or fgt_instnorm_apply_split for fnet), so the 3 x 3 layers run the LDS-DMA tap kernels instead of
        the register-staged one (64 -> 64 at 240x432: 190 -> ~300 TFLOP/s); a block's output is ALSO kept in fp32 where it is the next block's residual.
        The same products as feeding fp32 tensors to the bf16x3 kernel (tap-routed layers sum them in another order: equal to fp32 rounding)."""
        sc = ops.DEFAULT_CONV_PRECISION != "fp32"
        K = (lambda pc, form: dict(out_split=form, out_h=False, out_il=ops.split_il(pc.Cout))) if sc else (lambda pc, form: {})      # (interleaved hand-over where C % 32 == 0)
        # y / y32: the running activation for conv consumers (Split in the chain) and its fp32 form (the residual operand)
        if instance:
            y = ops.instnorm(ops.conv2d(x, P["conv1"], stride=2, pad=3), act="relu", out_split="both" if sc else None)
        else:
            y = ops.conv2d(x, P["conv1"], stride=2, pad=3, act="relu", **K(P["conv1"], "both"))
        y32, y = y if sc else (y, y)
        nb = len(P["blocks"])
        for i, b in enumerate(P["blocks"]):
            s = b["stride"]
            last = i + 1 == nb                         # the last block's output only feeds the 1 x 1 output conv
            both = ("only" if last else "both") if sc else None
            if instance:
                h = ops.instnorm(ops.conv2d(y, b["c1"], stride=s, pad=1), act="relu", out_split="only" if sc else None)
                sk = y32 if b["down"] is None else ops.instnorm(ops.conv2d(y, b["down"], stride=s, pad=0))
                o = ops.instnorm(ops.conv2d(h, b["c2"], stride=1, pad=1), act="relu", res=sk, act2="relu", out_split=both)
            else:
                h = ops.conv2d(y, b["c1"], stride=s, pad=1, act="relu", **K(b["c1"], "only"))
                sk = y32 if b["down"] is None else ops.conv2d(y, b["down"], stride=s, pad=0)
                o = ops.conv2d(h, b["c2"], stride=1, pad=1, act="relu", epi="add", aux1=sk, act2="relu", **K(b["c2"], both))
            if not sc:
                y32 = y = o
            elif last:
                y32, y = None, o
            else:
                y32, y = o
        return ops.conv2d(y, P["conv2"])

    # ------------------------------------------------------------------ forward
    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        with torch.no_grad():
            if self.use_graph and flow_init is None and image1.is_cuda:
                # ~40 launches per GRU iteration: replay the whole pair as one hipGraph (fgt_amd/graph.py)
                from .graph import GraphCache
                key = (iters, bool(test_mode))
                if key not in self._graphs:
                    self._graphs[key] = GraphCache(lambda a, b: self._forward(a, b, iters, None, test_mode),
                                                   state_key=lambda: tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers())))
                out = self._graphs[key](image1.float(), image2.float())
                return tuple(o.clone() for o in out) if isinstance(out, tuple) else [o.clone() for o in out]
            return self._forward(image1, image2, iters, flow_init, test_mode)

    def pack_images(self, images):
        """[B,3,H,W] in 0..255 -> channels-last 2*(x/255)-1 with a zero 4th channel (raft.py:90-94)."""
        B, _, H, W = images.shape
        if H % 8 or W % 8:
            raise ValueError("RAFT needs H, W divisible by 8 (use InputPadder as the reference does)")
        out = torch.empty(B, H, W, 4, dtype=torch.float32, device=images.device)
        ops.nchw_to_nhwc(images.float(), out, coff=0, zero_to=4, scale=2.0 / 255.0, shift=-1.0)
        return out

    def encode_features(self, packed):
        """fnet (InstanceNorm: per-sample, so any batching of frames is exact) -> [B, H/8, W/8, 256]."""
        return self._encode(packed, self.packed()["fnet"], instance=True)

    def encode_context(self, packed):
        """cnet (BatchNorm in eval mode) -> [B, H/8, W/8, 256] (tanh/relu split applied by iterate)."""
        return self._encode(packed, self.packed()["cnet"], instance=False)

    def _forward(self, image1, image2, iters, flow_init, test_mode):
        B = image1.shape[0]
        imgs = torch.cat([self.pack_images(image1), self.pack_images(image2)], 0)
        fmap = self.encode_features(imgs)                                          # [2B, h8, w8, 256]
        cmap = self.encode_context(imgs[:B])                                       # [B, h8, w8, 256]
        return self.iterate(fmap[:B], fmap[B:], cmap, iters, flow_init, test_mode)

    def iterate(self, fmap1, fmap2, cmap, iters=12, flow_init=None, test_mode=True):
        """Correlation volume + pyramid, then the GRU refinement loop (raft.py:102-143) from precomputed features."""
        P = self.packed()
        B, h8, w8, _ = fmap1.shape
        dev = fmap1.device
        # all-pairs correlation volume (corr.py:52-60) as a GEMM against fmap2, then the avg-pool pyramid
        n = h8 * w8
        vol = torch.empty(B * n, n, dtype=torch.float32, device=dev)
        if self.batched_corr and ops.DEFAULT_CONV_PRECISION != "fp32" and n % 8 == 0:
            # bf16x3: both feature maps split once (the same hi / lo values the per-pair GEMM derived from them), then ONE launch for the batch:
            # group b multiplies fmap1[b] with fmap2[b] as it lies (fgt_conv_desc.gb_*): no per-pair weight packing, B x the tiles per launch
            f1s = ops.split(fmap1.reshape(B * n, 256), interleave=True, h=False)
            f2s = ops.split(fmap2.reshape(B * n, 256), interleave=True, h=False)
            ops.batched_gemm_nt(f1s.view(B, n, 256), f2s.view(B, n, 256), vol.view(B, n, n), scale=1.0 / 16.0)
        else:
            for b in range(B):
                pc = PackedConv(fmap2[b].reshape(n, 256), None)
                ops.linear(fmap1[b].reshape(n, 256), pc, out=vol[b * n:(b + 1) * n], out_scale=1.0 / 16.0)
        pyr = [vol]
        hh, ww = h8, w8
        for _ in range(self.corr_levels - 1):
            pyr.append(ops.avgpool2(pyr[-1], B * n, hh, ww))
            hh, ww = hh // 2, ww // 2
        cmap = cmap.contiguous()
        net = ops.axpby(cmap.view(B * n, 256)[:, :128], act="tanh")                # raft.py:112-115
        xbuf = torch.empty(B * n, 256, dtype=torch.float32, device=dev)            # [inp | motion(126) | flow(2)]
        ops.axpby(cmap.view(B * n, 256)[:, 128:], act="relu", out=xbuf[:, :128])
        ys, xs = torch.meshgrid(torch.arange(h8, device=dev), torch.arange(w8, device=dev), indexing="ij")
        coords0 = torch.stack([xs, ys], -1).float().unsqueeze(0).repeat(B, 1, 1, 1).reshape(B * n, 2).contiguous()
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = ops.axpby(coords1, 1.0, flow_init.permute(0, 2, 3, 1).reshape(B * n, 2).contiguous().float(), 1.0)
        flow4 = torch.zeros(B * n, 4, dtype=torch.float32, device=dev)
        corr = torch.empty(B * n, 324, dtype=torch.float32, device=dev)
        E, G = P["enc"], P["gru"]
        m4 = lambda t2: t2.view(B, h8, w8, t2.shape[1]) if t2.is_contiguous() else t2.unflatten(0, (B, h8, w8))
        # bf16x3 arithmetic: the conv -> conv chains of the update block hand their activations over PRE-SPLIT (ops.Split, bf16 hi / lo pairs:
        # the consumer's im2col tiles are plain LDS-DMA copies, csrc/conv_split.hip) — the same PRODUCTS as feeding fp32 tensors to the
        # bf16x3 kernel; layers that the geometry routes to the tap-reusing kernel (3x3, 1x5, 5x1 'same' convs on split inputs, csrc/conv_taps.hip)
        # sum them in (ky, chunk, kx) order instead of (ky, kx, chunk), so the two input forms agree to fp32 rounding, not bit for bit
        # (bit-identity holds within one input form: across tiles, batch sizes, ranks).  Under the 'f16' mode RAFT stays in bf16x3 (h = False: bf16 pairs, never fp16 planes).
        sc = ops.DEFAULT_CONV_PRECISION != "fp32"
        rows = B * n
        S = lambda ch: ops.Split.empty((rows, ch), dev, h=False)
        s4 = lambda sp: sp.view(B, h8, w8, sp.shape[-1])
        if sc:
            cor1_s, cor2_s, flo1_s, flo2_s, rh_s, net_s, xbuf_s = S(256), S(192), S(128), S(64), S(128), S(128), S(256)
            corr_s = S(E["convc1p"].Cin)                                            # 324 taps + 28 zero channels
            ops.split(cmap.view(rows, 256)[:, 128:], relu=True, out=xbuf_s.channels(0, 128))       # inp = relu(cnet[:, 128:])
            ops.split(net, out=net_s)
            motion_s, flow_s = xbuf_s.channels(128, 256), xbuf_s.channels(254, 256)
        hoist = self.hoist_context
        zr = False
        pads = {"1": (0, 2), "2": (2, 0)}
        if hoist:
            # the context term of every GRU conv, once per pair: conv over inp + bias, fp32 maps [rows, 128] (update.py:45-58 restricted to hx[:, 128:256])
            GI, GC = P["gru_it"], P["gru_ctx"]
            inp_in = s4(xbuf_s.channels(0, 128)) if sc else m4(xbuf)[..., :128]
            zr = self.fuse_zr and sc
            ctx = {g_ + s_: ops.conv2d(inp_in, GC["conv" + g_ + s_], pad=pads[s_]) for s_ in ("1", "2") for g_ in (("zr", "q") if zr else ("z", "r", "q"))}
            x_it = s4(motion_s) if sc else m4(xbuf)[..., 128:]
        ups = []
        for it in range(iters):
            if sc:
                ops.corr_lookup(pyr, B, h8, w8, self.corr_radius, coords1, out_s=s4(corr_s))
            else:
                ops.corr_lookup(pyr, B, h8, w8, self.corr_radius, coords1, m4(corr))
            ops.axpby(coords1, 1.0, coords0, -1.0, out=flow4[:, :2])               # flow = coords1 - coords0
            if sc:
                # BasicMotionEncoder (update.py:62-76)
                ops.conv2d(s4(corr_s), E["convc1p"], act="relu", out_split="only", out_s=cor1_s)
                ops.conv2d(s4(cor1_s), E["convc2"], pad=1, act="relu", out_split="only", out_s=cor2_s)
                ops.conv2d(m4(flow4), E["convf1"], pad=3, act="relu", out_split="only", out_s=flo1_s)
                ops.conv2d(s4(flo1_s), E["convf2"], pad=1, act="relu", out_split="only", out_s=flo2_s)
                ops.conv2d(s4(cor2_s), E["conv128"], x1=s4(flo2_s), pad=1, act="relu", out_split="only", out_s=motion_s)
                ops.split(flow4[:, :2], out=flow_s)
                # SepConvGRU (update.py:36-60): horizontal then vertical pass; z stays fp32 (an epilogue operand), r * h goes on split, the new
                # hidden state is written in both forms (fp32: the next pass's epilogue operands and the heads; split: the next convs' input)
                for s_, pad in (("1", (0, 2)), ("2", (2, 0))):
                    if zr:
                        gq = (GI["convq" + s_], dict(x1=x_it, bias_map=ctx["q" + s_]))
                        z, _ = ops.conv2d(s4(net_s), GI["convzr" + s_], x1=x_it, bias_map=ctx["zr" + s_], pad=pad, act="sigmoid", epi="mul", aux1=net,
                                          out_split="both", out_s=s4(rh_s), dual=True)
                    else:
                        gz, gr, gq = ((GI["conv" + g_ + s_], dict(x1=x_it, bias_map=ctx[g_ + s_])) if hoist else (G["conv" + g_ + s_], dict(x1=s4(xbuf_s))) for g_ in "zrq")
                        z = ops.conv2d(s4(net_s), gz[0], pad=pad, act="sigmoid", **gz[1])
                        ops.conv2d(s4(net_s), gr[0], pad=pad, act="sigmoid", epi="mul", aux1=net, out_split="only", out_s=rh_s, **gr[1])
                    net, _ = ops.conv2d(s4(rh_s), gq[0], pad=pad, act="tanh", epi="gru", aux1=z, aux2=net, out_split="both", out_s=net_s, **gq[1])
                    net = net.view(rows, 128)
                d = ops.conv2d(s4(net_s), P["fh"][0], pad=1, act="relu")
            else:
                cor = ops.conv2d(m4(corr), E["convc1"], act="relu")
                cor = ops.conv2d(cor, E["convc2"], pad=1, act="relu")
                flo = ops.conv2d(m4(flow4), E["convf1"], pad=3, act="relu")
                flo = ops.conv2d(flo, E["convf2"], pad=1, act="relu")
                ops.conv2d(cor, E["conv"], x1=flo, pad=1, act="relu", out=m4(xbuf)[..., 128:254])
                ops.axpby(flow4[:, :2], out=xbuf[:, 254:256])
                for s_, pad in (("1", (0, 2)), ("2", (2, 0))):
                    gz, gr, gq = ((GI["conv" + g_ + s_], dict(x1=x_it, bias_map=ctx[g_ + s_])) if hoist else (G["conv" + g_ + s_], dict(x1=m4(xbuf))) for g_ in "zrq")
                    z = ops.conv2d(m4(net), gz[0], pad=pad, act="sigmoid", **gz[1])
                    rh = ops.conv2d(m4(net), gr[0], pad=pad, act="sigmoid", epi="mul", aux1=net, **gr[1])
                    net = ops.conv2d(rh, gq[0], pad=pad, act="tanh", epi="gru", aux1=z, aux2=net, **gq[1]).view(B * n, 128)
                d = ops.conv2d(m4(net), P["fh"][0], pad=1, act="relu")
            # FlowHead + coords update (update.py:6-15, raft.py:131-132)
            new_coords = torch.empty_like(coords1)
            ops.conv2d(d, P["fh"][1], pad=1, epi="add", aux1=coords1, out=m4(new_coords))
            coords1 = new_coords
            if (not test_mode) or it == iters - 1:                                  # only the last mask is consumed in test_mode
                if sc:
                    mk = ops.conv2d(s4(net_s), P["mask"][0], pad=1, act="relu", out_split="only", out_s=S(256))
                    mk = ops.conv2d(s4(mk), P["mask"][1], out_scale=0.25)
                else:
                    mk = ops.conv2d(m4(net), P["mask"][0], pad=1, act="relu")
                    mk = ops.conv2d(mk, P["mask"][1], out_scale=0.25)
                ops.axpby(coords1, 1.0, coords0, -1.0, out=flow4[:, :2])
                ups.append(ops.convex_upsample(m4(flow4), mk))
        if test_mode:
            flow_low = ops.nhwc_to_nchw(m4(ops.axpby(coords1, 1.0, coords0, -1.0)))
            return flow_low, ups[-1]
        return ups
