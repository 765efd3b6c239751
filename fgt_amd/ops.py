"""Tensor-level wrappers over the C ABI (include/fgt_hip.h).

PyTorch is used here only for device memory (torch.empty through the caching allocator), the current HIP
stream and one-time weight re-layout; every computation on activations is a libfgt_hip.so kernel.
Activations are channels-last fp32 tensors [N, H, W, C] (or [rows, C] token matrices); channel slices of wider
buffers are ordinary torch views (stride(-1) == 1, pixel stride = stride(-2)), so concatenations are never copied.
"""
import ctypes as C
import math
import os
import warnings

import torch

from . import _lib
from ._lib import ACT, EPI, PREC, TILE, AttnDesc, ConvDesc, check

# arithmetic mode of fgt_conv2d when the caller does not pass `precision=`: 'fp32' (exact), 'bf16x3' (16 significant bits per operand,
# 3 MFMAs per product) or 'f16' (operands rounded once to fp16 by their producer — 11 bits, one MFMA per product, half the activation
# bytes; GEMMs whose input is still an fp32 tensor run in bf16x3)
DEFAULT_CONV_PRECISION = os.environ.get("FGT_CONV_PRECISION", "fp32")
DEFAULT_ATTN_PRECISION = os.environ.get("FGT_ATTN_PRECISION", "fp32")

# Per-shape tile autotuning of fgt_conv2d: the first call of a new (shape, precision) times every tile candidate with HIP
# events and caches the fastest.  Tiles only change the work decomposition: within ONE kernel family (tap-reusing kernel / the
# others) results are bit-identical across tiles (the k order of every accumulation is the same), and which family a layer gets is
# decided by its geometry (fgt_conv_taps_route), never by tuning — so tuning never changes numerics.  (The two families accumulate in
# different orders — (ky, chunk, kx) vs (ky, kx, chunk) — so a split-input layer routed to the tap kernel differs in the last bits from
# the same layer fed fp32 tensors; same products, different summation order.)
AUTOTUNE = os.environ.get("FGT_AUTOTUNE", "1") != "0"
TILE_CANDIDATES = ("128x128", "64x64", "128x64", "256x128", "128x32", "128x128x8", "256x128x16", "256x64x8",
                   "c4",    # 4-channel fp32 inputs only (csrc/conv_c4.hip; rejected, hence skipped, elsewhere): bit-identical to the tiles above
                   # split inputs only (rejected, hence skipped, for fp32 inputs): the same tiles with early stage release
                   "128x128ea", "64x64ea", "128x64ea", "128x128x8ea", "256x128x16ea", "256x64x8ea",
                   # fp16 kernel only (csrc/conv_f16.hip): one more early-release tile, and every tile on the wide LDS image
                   "256x128ea", "128x128w", "64x64w", "128x64w", "256x128w", "128x32w", "128x128x8w", "256x128x16w", "256x64x8w",
                   "128x128eaw", "64x64eaw", "128x64eaw", "128x128x8eaw", "256x128x16eaw", "256x64x8eaw", "256x128eaw")
# Layers that fgt_conv2d routes to the tap-reusing kernel (csrc/conv_taps.hip; decided by geometry: fgt_conv_taps_route) are tuned among ITS
# tiles only — they are bit-identical to each other, so results never depend on tuning.
TAPS_CANDIDATES = ("128x128x8t", "128x128t", "128x64t", "128x64x8t", "64x64t", "128x128it", "256x128it", "256x256it")     # (...it: csrc/conv_taps_il.hip, interleaved requests; declines k x 1 / upsampling layers)
_FORCE_TILE_ORDER = int(os.environ["FGT_CONV_TILE_ORDER"]) if os.environ.get("FGT_CONV_TILE_ORDER") else None      # A/B: fgt_conv_desc.tile_order for every layer
_tile_cache = {}
_tile_validated = set()      # keys whose cached tile has been checked against the geometry's kernel family (conv2d)


def mode_key():
    """Everything outside the tensors that changes what a launch sequence computes: graph caches key on it (fgt_amd/graph.py)."""
    return (DEFAULT_CONV_PRECISION, DEFAULT_ATTN_PRECISION, WEIGHTS_INTERLEAVED, SPLIT_INTERLEAVED)


def tuning_table():
    return {repr(k): v for k, v in _tile_cache.items()}


def save_tuning(path):
    import json
    with open(path, "w") as f:
        json.dump(tuning_table(), f, indent=0, sort_keys=True)


def load_tuning(path):
    """Pre-seed the autotuner with a table written by save_tuning (keys are shape tuples, values tile codes)."""
    import ast
    import json
    for k, v in json.load(open(path)).items():
        key = ast.literal_eval(k)
        _tile_cache[key] = int(v)
        _tile_validated.discard(key)            # a loaded tile is checked against the geometry's kernel family again (conv2d)


if os.environ.get("FGT_TUNING_FILE") and os.path.exists(os.environ["FGT_TUNING_FILE"]):
    load_tuning(os.environ["FGT_TUNING_FILE"])
if os.environ.get("FGT_TUNING_FILE") and os.environ.get("FGT_TUNING_SAVE") == "1":      # tools: write the table this process tuned when it exits
    import atexit
    atexit.register(lambda: save_tuning(os.environ["FGT_TUNING_FILE"]))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _require_dev(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise RuntimeError("fgt_amd ops need tensors on the MI355X (cuda) device; there is no CPU path")
            _lib.init_device(t.device.index)
            if t.dtype != torch.float32:
                raise RuntimeError(f"fgt_amd ops compute in fp32, got {t.dtype} (split bf16 operands travel as ops.Split)")


class Split:
    """A "split" activation tensor (include/fgt_hip.h, fgt_conv_desc.in_split): the bf16 pair hi = bf16_rne(x),
    lo = bf16_rne(x - hi) of the fp32 tensor `shape` it stands for (same bytes as fp32) — or, in the 'f16' mode (h = True), the ONE
    fp16 plane data = f16_rne(x) [*shape] (half the bytes; fgt_conv_desc.in_split = 3).  Two bf16 layouts:
      planes       data = bf16 [2, *shape]: data[0] = hi, data[1] = lo (any channel count that is a multiple of 8);
      interleaved  data = bf16 [*shape[:-1], 2*C]: per 32 channels [hi 32 | lo 32], so the 64 + 64 bytes one K-step of the conv
                   kernel needs from a pixel are ONE 128-byte line (C % 32 == 0).
    Produced once by the kernel that computes x (conv epilogue or ops.split), consumed by fgt_conv2d's LDS-DMA loader."""

    def __init__(self, data, interleaved=False, h=False):
        if h:
            assert data.dtype == torch.float16 and data.is_cuda and not interleaved
        else:
            assert data.dtype == torch.bfloat16 and data.is_cuda and (interleaved or data.shape[0] == 2)
        self.data, self.il, self.h = data, interleaved, h

    @staticmethod
    def empty(shape, device, interleaved=False, h=None):
        """h: the fp16 format (one plane); None = follow the arithmetic mode (DEFAULT_CONV_PRECISION == 'f16')."""
        if h is None:
            h = DEFAULT_CONV_PRECISION == "f16" and not interleaved
        if h:
            return Split(torch.empty(tuple(shape), dtype=torch.float16, device=device), h=True)
        if interleaved:
            assert shape[-1] % 32 == 0, "interleaved split tensors need C % 32 == 0"
            return Split(torch.empty(tuple(shape[:-1]) + (2 * shape[-1],), dtype=torch.bfloat16, device=device), True)
        return Split(torch.empty((2,) + tuple(shape), dtype=torch.bfloat16, device=device))

    @property
    def shape(self):
        if self.h:
            return self.data.shape
        return torch.Size(tuple(self.data.shape[:-1]) + (self.data.shape[-1] // 2,)) if self.il else self.data.shape[1:]

    @property
    def device(self):
        return self.data.device

    @property
    def hi(self):
        """The tensor whose data_ptr / strides describe the hi values (planes: plane 0; interleaved: the 2*C-wide rows)."""
        return self.data if (self.il or self.h) else self.data[0]

    @property
    def ps(self):
        """Plane stride argument of the C ABI: bf16 elements between hi and lo, 32 = interleaved, -1 = one fp16 plane."""
        return -1 if self.h else (32 if self.il else self.data.stride(0))

    def view(self, *shape):
        if self.h:
            return Split(self.data.view(*shape), h=True)
        return Split(self.data.view(*shape[:-1], 2 * shape[-1] if shape[-1] >= 0 else -1), True) if self.il else Split(self.data.view(2, *shape))

    def __getitem__(self, idx):            # leading-dimension slices (rows / frames)
        if self.h:
            return Split(self.data[idx], h=True)
        return Split(self.data[idx], True) if self.il else Split(self.data[:, idx])

    def channels(self, c0, c1):
        """Channel slice [c0, c1) of the last dimension as a Split view of the same storage (writers fill slices of a wider buffer in
        place: RAFT's GRU input [inp | motion | flow]); interleaved: multiples of 32."""
        if self.h:
            return Split(self.data[..., c0:c1], h=True)
        if self.il:
            assert c0 % 32 == 0 and c1 % 32 == 0
            return Split(self.data[..., 2 * c0:2 * c1], True)
        return Split(self.data[..., c0:c1])

    def planes(self):
        """(hi, lo) as bf16 tensors of the logical shape (tests / debugging)."""
        assert not self.h, "an fp16 Split has one plane"
        if not self.il:
            return self.data[0], self.data[1]
        d = self.data.reshape(*self.data.shape[:-1], -1, 2, 32)
        return d[..., 0, :].reshape(self.shape), d[..., 1, :].reshape(self.shape)

    def float(self):
        """hi + lo as fp32 (tests / debugging: 16 mantissa bits of the original)."""
        if self.h:
            return self.data.float()
        hi, lo = self.planes()
        return hi.float() + lo.float()


def split(x, relu=False, out=None, interleave=False, h=None):
    """fp32 [rows, C] / [N,H,W,C] -> Split (fgt_split); the format follows the arithmetic mode unless `out` or `h` (True: one fp16
    plane, False: the bf16 pair) is given."""
    _require_dev(x)
    x4, N, H, W, Cc, ld = _as_map(x)
    if out is None:
        out = Split.empty(x.shape, x.device, interleave, h=h)
    o4, oN, oH, oW, oC, ldo = _as_map(out.hi)
    assert (oN * oH * oW, oC) == (N * H * W, Cc * (2 if out.il else 1))
    check(_lib.lib().fgt_split(_ptr(x4), N * H * W, Cc, ld, _ptr(out.data), ldo, out.ps, int(relu), _stream()), "fgt_split")
    return out


def _as_map(x):
    """[N,H,W,C] (or [rows,C]) view -> (x4, N, H, W, C, pixel stride) after validating channels-last strides."""
    if x.dim() == 2:
        x = x.unsqueeze(0).unsqueeze(0)
    if x.dim() != 4:
        raise RuntimeError(f"expected a [N,H,W,C] or [rows,C] tensor, got shape {tuple(x.shape)}")
    N, H, W, Cc = x.shape
    if Cc > 1 and x.stride(3) != 1:
        raise RuntimeError("activation must be channels-last (stride(-1) == 1)")
    if W > 1:
        ld = x.stride(2)
    elif H > 1:
        ld = x.stride(1)
    elif N > 1:
        ld = x.stride(0)
    else:
        ld = Cc
    if (H > 1 and W > 1 and x.stride(1) != W * ld) or (N > 1 and H * W > 1 and x.stride(0) != H * W * ld):
        raise RuntimeError(f"activation pixels must be uniformly strided: shape {tuple(x.shape)} strides {x.stride()}")
    return x, N, H, W, Cc, ld


def ceil_to(a, m):
    return (a + m - 1) // m * m


class PackedConv:
    """Weights of one conv / linear in the kernel's layout: [groups, Npad, Kpad], k = (ky*kw+kx)*Cg + ci."""

    def __init__(self, w, bias=None, groups=1, scale=None, pad_cin_to4=True):
        if w.dim() == 2:
            w = w[:, :, None, None]
        Cout, Cg, kh, kw = w.shape
        assert Cout % groups == 0
        self.Cout, self.groups, self.kh, self.kw = Cout, groups, kh, kw
        Cg_p = ceil_to(Cg, 4) if pad_cin_to4 else Cg
        self.Cg = Cg_p
        self.k_alg = kh * kw * Cg                    # algorithmic K (before channel padding): roofline accounting
        self.Cin = Cg_p * groups
        K = kh * kw * Cg_p
        self.K = K
        self.Kpad = ceil_to(K, 32)
        Cout_g = Cout // groups
        self.Npad = ceil_to(Cout_g, 128)
        wp = w.detach().float().reshape(groups, Cout_g, Cg, kh, kw).permute(0, 1, 3, 4, 2)  # [G, Cout_g, kh, kw, Cg]
        packed = torch.zeros(groups, self.Npad, self.Kpad, dtype=torch.float32, device=w.device)
        tmp = torch.zeros(groups, Cout_g, kh, kw, Cg_p, dtype=torch.float32, device=w.device)
        tmp[..., :Cg] = wp
        packed[:, :Cout_g, :K] = tmp.reshape(groups, Cout_g, K)
        self.w = packed.contiguous()
        self._w_split = None
        self.bias = None if bias is None else bias.detach().float().contiguous()
        self.scale = None if scale is None else scale.detach().float().contiguous()


WEIGHTS_INTERLEAVED = os.environ.get("FGT_W_IL", "1") != "0"
# bf16x3 mode: split tensors whose producer can write the interleaved layout ([hi 32 | lo 32] per 32 channels: conv epilogue, fgt_split,
# fgt_layernorm, fgt_fold) and whose consumers are convs / GEMMs use it, so that the consumer can run on the wide LDS image (csrc/conv_wide.hip:
# 8-row x 128-byte LDS-DMA pieces; the autotuner picks per shape between it and the 16-row x 64-byte kernel).  Same values, same products:
# results are bit-identical either way.  FGT_SPLIT_IL=0: planes everywhere (A/B measurements).
SPLIT_INTERLEAVED = os.environ.get("FGT_SPLIT_IL", "1") != "0"


def split_il(channels):
    """Should a new split tensor of `channels` channels be interleaved?  (bf16 pairs only: the 'f16' mode has its own single-plane format)"""
    return SPLIT_INTERLEAVED and DEFAULT_CONV_PRECISION == "bf16x3" and channels % 32 == 0


def _f16_weights(pc):
    """The fp16 kernel's weight image: f16_rne(w) as [groups, Npad, Kpad64], K zero-padded to the 64-channel K-step of csrc/conv_f16.hip."""
    cache = pc.__dict__.setdefault("_w_split_cache", {})
    if "h" not in cache:
        G, Np, Kp = pc.w.shape
        Kp64 = ceil_to(Kp, 64)
        w = torch.zeros(G, Np, Kp64, dtype=torch.float16, device=pc.w.device)
        w[:, :, :Kp] = pc.w.clamp(-65504.0, 65504.0).to(torch.float16)
        cache["h"] = w
    return cache["h"]


def _split_weights(pc, interleaved=None):
    """The bf16x3 kernels' weight image: hi = bf16_rne(w), lo = bf16_rne(w - hi), as two planes [2, groups, Npad, Kpad] or
    (fgt_conv_desc.w_il) interleaved per K-step [groups, Npad, Kpad/32, (hi 32 | lo 32)] — one 128-byte line per row and step."""
    il = WEIGHTS_INTERLEAVED if interleaved is None else interleaved
    cache = pc.__dict__.setdefault("_w_split_cache", {})
    if il not in cache:
        hi = pc.w.to(torch.bfloat16)
        lo = (pc.w - hi.float()).to(torch.bfloat16)
        if il:
            G, Np, Kp = hi.shape
            cache[il] = torch.stack([hi.view(G, Np, Kp // 32, 32), lo.view(G, Np, Kp // 32, 32)], 3).contiguous()
        else:
            cache[il] = torch.stack([hi, lo], 0).contiguous()
    return cache[il], il


def _frag_weights(pc):
    """The bf16 hi/lo weight image in MFMA fragment order (fgt_conv_desc.w_il = 2, diagnostic builds): [groups, Kpad/32, Npad/32, hi | lo,
    k-half, 64 lanes] 16 bytes, lane = (k / 8 % 2) * 32 + channel-out % 32 — a wave-wide 16-byte load is 1 KB of consecutive memory."""
    cache = pc.__dict__.setdefault("_w_split_cache", {})
    if "frag" not in cache:
        il, _ = _split_weights(pc, interleaved=True)                       # [G, Npad, Kpad/32, 2, 32]
        G, Np, KS = il.shape[:3]
        cache["frag"] = il.view(G, Np // 32, 32, KS, 2, 2, 2, 8).permute(0, 3, 1, 4, 5, 6, 2, 7).contiguous()
    return cache["frag"]


def prepack_weights(pc, mode=None):
    """Build the weight images fgt_conv2d would otherwise create lazily on its first launch in arithmetic `mode` (default: the
    current one): the bf16 hi/lo image for 'bf16x3' (and for the GEMMs that still take fp32 inputs in the 'f16' mode) and the fp16
    image for 'f16'.  Called on the main stream before work is forked onto side streams (scheduler.ClipRunner)."""
    mode = DEFAULT_CONV_PRECISION if mode is None else mode
    if pc.Cout // pc.groups <= 4:
        return
    if mode in ("bf16x3", "f16"):
        _split_weights(pc)
    if mode == "f16":
        _f16_weights(pc)


# "nearest x2 + 3x3" as a 2x2 convolution with a sub-pixel output (fgt_conv_desc.ps_phase_pad, ABI 9): 4 multiply-adds per output value and input
# channel instead of 9.  FGT_UP4=0: the upsampled 3x3 form on the tap kernels (A/B measurements; results differ by the rounding of the weight sums).
UP4 = os.environ.get("FGT_UP4", "1") != "0"


def up4_weights(w, cpad=None):
    """[Cout, Cin, 3, 3] -> [4*cpad, Cin, 2, 2] (cpad >= Cout, default Cout: zero rows pad a sub-pixel's channels), rows ordered (a, b, co): sub-pixel (a, b) of the x2 output is a 2x2 convolution over the
    low-resolution map whose taps are sums of the 3x3 taps that read the same input pixel — rows: a = 0 -> [w(-1)], [w(0) + w(+1)] over input
    rows {i-1, i}; a = 1 -> [w(-1) + w(0)], [w(+1)] over {i, i+1}; columns likewise (network_blocks_2d.py:46-60: F.interpolate(scale_factor=2)
    + 3x3 conv, padding 1).  Sums in fp64, rounded once."""
    w = w.detach().double()
    Cout, Cin = w.shape[:2]
    rows = (lambda t: (t[:, :, 0], t[:, :, 1] + t[:, :, 2]), lambda t: (t[:, :, 0] + t[:, :, 1], t[:, :, 2]))      # over dim 2 (ky)
    cpad = Cout if cpad is None else cpad
    out = torch.zeros(2, 2, cpad, Cin, 2, 2, dtype=torch.float64, device=w.device)
    for a in range(2):
        for ky, wr in enumerate(rows[a](w)):                        # wr [Cout, Cin, 3 (kx)]
            for b in range(2):
                cols = (wr[:, :, 0], wr[:, :, 1] + wr[:, :, 2]) if b == 0 else (wr[:, :, 0] + wr[:, :, 1], wr[:, :, 2])
                for kx, wc in enumerate(cols):
                    out[a, b, :Cout, :, ky, kx] = wc
    return out.reshape(4 * cpad, Cin, 2, 2).float()


def _up4_pack(pc):
    """The PackedConv of the 2x2 sub-pixel form of a packed 3x3 layer (cached on it)."""
    cache = pc.__dict__.setdefault("_w_split_cache", {})
    if "up4" not in cache:
        w = pc.w[0, :pc.Cout, :pc.K].reshape(pc.Cout, 3, 3, pc.Cg).permute(0, 3, 1, 2)      # packed k = (ky*3 + kx)*Cg + ci -> [Cout, Cg, 3, 3]
        cp = ceil_to(pc.Cout, 64)                                   # a sub-pixel's columns padded to the tile width (LAFC: 48 -> 64, 96 -> 128)
        rep = lambda v, fill: None if v is None else torch.cat([v, v.new_full((cp - pc.Cout,), fill)]).repeat(4)
        q = PackedConv(up4_weights(w, cp), rep(pc.bias, 0.0), scale=rep(pc.scale, 1.0), pad_cin_to4=False)
        q.k_alg, q.up4_c = 4 * (pc.k_alg // 9), cp
        cache["up4"] = q
    return cache["up4"]


def prepack_up4(pc):
    """Build the 2x2 sub-pixel weight image of a 3x3 layer that is used with upsample=True (and its bf16 hi / lo image) on the CURRENT stream:
    for callers that fork work onto side streams or capture graphs (fgt_model.Model.prepack)."""
    if UP4 and pc.kh == 3 and pc.kw == 3 and pc.groups == 1 and pc.Cout % 4 == 0 and pc.Cout >= 32 and pc.Cg % 32 == 0:
        _split_weights(_up4_pack(pc))


def _up4_ok(x, pc, stride, pad, dil, pad_mode, in_relu, epi, out_nchw, tile, precision, ps, ky_skip_n0, aux_per_image, bias_map, dual, out_s, out_il, out_split):
    return (UP4 and "_up4_declined" not in pc.__dict__ and isinstance(x, Split) and not x.h and pc.kh == 3 and pc.kw == 3 and pc.groups == 1 and pc.Cout % 4 == 0 and pc.Cout >= 32 and stride == 1 and pad == 1 and dil == 1 and
            pad_mode == "zeros" and not in_relu and epi in (None, "mul", "add") and not out_nchw and tile is None and ps is None and not ky_skip_n0 and
            not aux_per_image and bias_map is None and not dual and (precision or DEFAULT_CONV_PRECISION) == "bf16x3" and pc.Cg % 32 == 0 and
            not (out_split and (out_il or (out_s is not None and out_s.il)) and pc.Cout % 32))



def conv2d(x, pc, x1=None, stride=1, pad=0, dil=1, upsample=False, pad_mode="zeros", in_relu=False, act=None, slope=0.2,
           epi=None, aux1=None, aux2=None, act2=None, out_scale=1.0, out=None, out_nchw=False, tile=None, precision=None,
           out_split=None, out_s=None, out_il=False, out_h=None, ps=None, ky_skip_n0=0, aux_per_image=False, n_alg=0, bias_map=None, tile_order=0,
           dual=False, _phase_pad=0):
    """fgt_conv2d.  x (and optional x1) are channels-last maps (fp32 tensors, or `Split`s for the LDS-DMA bf16x3 path);
    returns/outputs a channels-last map (or NCHW).  out_split: None -> fp32 result; "only" -> a Split; "both" -> (fp32, Split).
    ps = (r, c, g0, Hf, Wf): sub-pixel output (fold as a convolution, fgt_conv_desc.ps_r): the result is the [N, Hf, Wf, c] map;
    ky_skip_n0 / aux_per_image / n_alg: the fields of the same name.  bias_map: an fp32 [N, Ho, Wo, Cout] (or [rows, Cout]) map added in front of
    the activation INSTEAD of pc.bias (fgt_conv_desc.ld_bias; the caller folds the bias into it).
    dual=True (fgt_conv_desc.dual_n0 = Cout / 2, with epi="mul", out_split="both"): two heads — columns [0, Cout/2) -> act(v) into the fp32 `out`
    [.., Cout/2]; columns [Cout/2, Cout) -> act(v) * aux1 into the Split `out_s` [.., Cout/2]."""
    if upsample and _up4_ok(x, pc, stride, pad, dil, pad_mode, in_relu, epi, out_nchw, tile, precision, ps, ky_skip_n0, aux_per_image, bias_map, dual, out_s, out_il, out_split):
        # nearest x2 + 3x3 -> the 2x2 sub-pixel form over the low-resolution map (4 of the 9 multiply-adds; ABI 9 ps_phase_pad)
        Hx, Wx = x.shape[-3], x.shape[-2]
        q = _up4_pack(pc)
        try:
            return conv2d(x, q, x1=x1, stride=1, pad=1, act=act, slope=slope, epi=epi, aux1=aux1, act2=act2, out_scale=out_scale, out=out,
                          out_split=out_split, out_s=out_s, out_il=out_il, out_h=out_h, ps=(2, q.up4_c, 2 * q.up4_c, 2 * Hx, 2 * Wx), n_alg=4 * pc.Cout, precision="bf16x3", _phase_pad=pc.Cout)
        except RuntimeError as e:
            # a geometry the 2x2 form's kernels decline (argument validation: nothing was launched): the upsampled 3x3 form serves every layer it served before
            pc.__dict__["_up4_declined"] = str(e)
            warnings.warn(f"conv2d: the 2x2 sub-pixel form declined this upsampled layer ({e}); using the upsampled 3x3 form")
    in_split = isinstance(x, Split)
    if in_split:
        assert x1 is None or (isinstance(x1, Split) and x1.il == x.il and x1.h == x.h), "conv2d: both sources must be split the same way"
        xs, x1s = x, x1
        x, x1 = xs.hi, (None if x1s is None else x1s.hi)
    else:
        _require_dev(x, x1)
    _require_dev(aux1, aux2, out, bias_map)
    x, N, H, W, C0, ld0 = _as_map(x)
    C1, ld1 = 0, 0
    if x1 is not None:
        x1, N1, H1, W1, C1, ld1 = _as_map(x1)
        assert (N1, H1, W1) == (N, H, W), "conv2d: sources differ in geometry"
    if in_split and xs.il:
        C0, C1 = C0 // 2, C1 // 2                    # interleaved rows hold 2*C elements
    assert C0 + C1 == pc.Cin, f"conv2d: input channels {C0}+{C1} != packed {pc.Cin}"
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    dh, dw = (dil, dil) if isinstance(dil, int) else dil
    Hin, Win = H * (2 if upsample else 1), W * (2 if upsample else 1)
    Ho = (Hin + 2 * ph - dh * (pc.kh - 1) - 1) // sh + 1
    Wo = (Win + 2 * pw - dw * (pc.kw - 1) - 1) // sw + 1
    if _phase_pad:
        Ho, Wo = H, W                            # one tile row per INPUT pixel; padding (1 - a, 1 - b) per sub-pixel (fgt_conv_desc.ps_phase_pad)
    osp = {None: 0, False: 0, "only": 1, "both": 2}[out_split]
    oshape = (N, Ho, Wo, pc.Cout) if ps is None else (N, ps[3], ps[4], _phase_pad or ps[1])       # (sub-pixel output: the folded map; ps_phase_pad: its real channels)
    if dual:
        assert epi == "mul" and out_split == "both" and ps is None and pc.groups == 1 and pc.Cout % 8 == 0, "conv2d: dual needs epi='mul', out_split='both', groups = 1"
        oshape = (N, Ho, Wo, pc.Cout // 2)
    if out is None and osp != 1:
        out = torch.empty((N, pc.Cout, Ho, Wo) if out_nchw else oshape, dtype=torch.float32, device=x.device)
    ldo = 0
    if out_nchw:
        assert ps is None and out.is_contiguous() and tuple(out.shape) == (N, pc.Cout, Ho, Wo)
    elif out is not None:
        o4, oN, oH, oW, oC, ldo = _as_map(out)
        assert (oN * oH * oW, oC) == (oshape[0] * oshape[1] * oshape[2], oshape[3]), f"conv2d: out shape {tuple(out.shape)} != {oshape}"
    d = ConvDesc()
    d.N, d.H, d.W = N, H, W
    d.C0, d.ld0, d.off0 = C0, ld0, 0
    d.C1, d.ld1, d.off1 = C1, ld1, 0
    d.Cout, d.groups = pc.Cout, pc.groups
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = pc.kh, pc.kw, sh, sw, ph, pw, dh, dw
    d.upsample, d.pad_mode, d.in_relu = int(upsample), {"zeros": 0, "replicate": 1}[pad_mode], int(in_relu)
    d.Ho, d.Wo, d.ldo, d.ooff, d.out_nchw = Ho, Wo, ldo, 0, int(out_nchw)
    d.act, d.slope, d.epi, d.act2 = ACT[act], float(slope), EPI[epi], ACT[act2]
    d.ld_aux1 = 0 if aux1 is None else _as_map(aux1)[5]
    d.ld_aux2 = 0 if aux2 is None else _as_map(aux2)[5]
    d.out_scale = float(out_scale)
    d.Kpad, d.Npad, d.tile = pc.Kpad, pc.Npad, TILE[tile]
    d.k_alg = getattr(pc, "k_alg", 0)
    if ps is not None:
        d.ps_r, d.ps_c, d.ps_g0, d.ps_H, d.ps_W = (int(v) for v in ps)
    d.ky_skip_n0, d.aux_per_image, d.n_alg = int(ky_skip_n0), int(bool(aux_per_image)), int(n_alg)
    d.ps_phase_pad = int(_phase_pad)
    # the epilogue indexes its operands by tile row with no bounds of their own: a table of another grid (a stale cache key, another th / tw) would
    # read out of bounds on the device
    rows_out = N * Ho * Wo
    for nm, a in (("aux1", aux1), ("aux2", aux2)):
        if a is None:
            continue
        a4, aN, aH, aW, aC, _ = _as_map(a)
        if nm == "aux2" and epi == "ps_add2":
            assert ps is not None and (aN * aH * aW, aC) == (N * ps[3] * ps[4], ps[1]), f"conv2d: ps_add2 aux2 shape {tuple(a.shape)} is not the [N, {ps[3]}, {ps[4]}, {ps[1]}] map"
        elif _phase_pad:
            assert (aN * aH * aW, aC) == (N * ps[3] * ps[4], _phase_pad), f"conv2d: {nm} shape {tuple(a.shape)} is not the [N, {ps[3]}, {ps[4]}, {_phase_pad}] output map"
        elif aux_per_image and (nm == "aux1" or epi == "affine"):
            assert (aN * aH * aW, aC) == (Ho * Wo, pc.Cout), f"conv2d: per-image {nm} table shape {tuple(a.shape)} != ({Ho * Wo}, {pc.Cout})"
        else:
            assert aN * aH * aW >= rows_out and aC >= (pc.Cout // 2 if dual else pc.Cout), f"conv2d: {nm} shape {tuple(a.shape)} is smaller than the output ({rows_out}, {pc.Cout})"
    d.tile_order = int(tile_order) if _FORCE_TILE_ORDER is None else _FORCE_TILE_ORDER
    d.dual_n0 = pc.Cout // 2 if dual else 0
    bias = pc.bias
    if bias_map is not None:
        b4, bN, bH, bW, bC, d.ld_bias = _as_map(bias_map)
        assert (bN * bH * bW, bC) == (N * Ho * Wo, pc.Cout), f"conv2d: bias_map shape {tuple(bias_map.shape)}"
        bias = b4
    prec = precision if precision is not None else DEFAULT_CONV_PRECISION
    if prec == "f16" and not (in_split and xs.h):
        prec = "bf16x3"                      # 'f16' = fp16 where the operand arrives as fp16; fp32 inputs are not rounded to 11 bits here
    d.precision = PREC[prec]
    if pc.Cout // pc.groups <= 4 and d.tile == 0 and not in_split and not osp:
        d.precision = 0                      # Cout <= 4 layers run the fp32 VALU direct-conv kernels
    d.in_split = (3 if xs.h else (2 if xs.il else 1)) if in_split else 0
    if in_split:
        if xs.h:
            d.precision = PREC["f16"]        # the format decides: fp16 tensors feed the fp16 kernel
        elif d.precision != PREC["bf16x3"]:
            raise RuntimeError("conv2d: Split inputs need precision='bf16x3'")
        d.ps0, d.ps1 = (0, 0) if xs.h else (xs.ps, (0 if x1s is None else x1s.ps))
    d.out_split = osp
    if osp:
        if out_s is None:
            # out_h: None = the split output follows the arithmetic mode (fp16 plane in 'f16'), False = always the bf16 pair (RAFT / LAFC
            # stay in bf16x3 under the 'f16' switch)
            out_s = Split.empty(oshape, x.device, interleaved=bool(out_il), h=False if out_il else out_h)
        s4, sN, sH, sW, sC, ldo_s = _as_map(out_s.hi)
        assert (sN * sH * sW, sC) == (oshape[0] * oshape[1] * oshape[2], oshape[3] * (2 if out_s.il else 1)), f"conv2d: out_s shape {tuple(out_s.shape)}"
        d.ldo_s, d.ooff_s, d.pso = ldo_s, 0, out_s.ps
    if d.precision == 0 or (in_split and xs.h and pc.Cout // pc.groups <= 4 and d.tile == 0):
        wbuf = pc.w                          # (fp16 map into a Cout <= 4 layer: fp32 weights and arithmetic, csrc/conv_direct.hip)
    elif d.precision == PREC["f16"]:
        wbuf = _f16_weights(pc)
        d.Kpad = wbuf.shape[-1]
    else:
        wbuf, wil = _split_weights(pc)
        d.w_il = int(wil)
        if d.tile >= 300:                    # diagnostic builds: weights in MFMA fragment order (csrc/diag/conv_taps_breg.hip)
            wbuf, d.w_il = _frag_weights(pc), 2
    args = (C.byref(d), _ptr(x), _ptr(x1), _ptr(wbuf), _ptr(pc.scale), _ptr(bias), _ptr(aux1), _ptr(aux2), _ptr(out),
            _ptr(None if out_s is None else out_s.data))
    if d.tile == 0 and AUTOTUNE and pc.Cout // pc.groups > 4:
        key = (N, H, W, C0, C1, pc.Cout, pc.groups, pc.kh, pc.kw, sh, sw, ph, pw, dh, dw, d.upsample, d.epi, d.out_nchw, d.precision,
               d.in_split, d.out_split, d.w_il, int(bool(osp) and out_s.il) + 2 * int(bool(osp) and out_s.h), d.pad_mode, d.in_relu,
               d.ps_r, d.ps_c, d.ps_H, d.ps_W, d.ky_skip_n0, int(d.ld_bias > 0), d.tile_order, d.aux_per_image, d.dual_n0 + 100000 * d.ps_phase_pad)
        best = _tile_cache.get(key)
        if best is not None and key not in _tile_validated:
            # A cached / loaded tile must belong to the kernel family the GEOMETRY routes this layer to (taps-routed layers: codes 200-299,
            # everything else below 200; diagnostic codes >= 300 never come from the autotuner): a table saved under FGT_CONV_TAPS=0 or from a
            # diagnostic build would otherwise select another accumulation order — or a tile the product library does not have.
            taps = bool(_lib.lib().fgt_conv_taps_route(C.byref(d)))
            if (200 <= best < 300) != taps or best >= 300 or best < 0:
                best = None
                _tile_cache.pop(key, None)
            else:
                _tile_validated.add(key)
        if best is None and not torch.cuda.is_current_stream_capturing() and not _aliases(out, out_s, x, x1, aux1, aux2, bias_map):
            # (tuning re-launches the kernel into the caller's buffers and synchronises: illegal under stream capture, and it would
            #  corrupt an output that aliases an input / aux operand — such calls run on the static tile and are not cached)
            taps = bool(_lib.lib().fgt_conv_taps_route(C.byref(d)))
            best = _tile_cache[key] = _autotune(d, args, TAPS_CANDIDATES if taps else TILE_CANDIDATES)
            _tile_validated.add(key)
        d.tile = best or 0
        if best:
            rc = _lib.lib().fgt_conv2d(*args, _stream())
            if rc != 0:                      # e.g. a table from a diagnostic build names a tile this library does not have: forget it, static tile
                _tile_cache.pop(key, None)
                _tile_validated.discard(key)
                d.tile = 0
                check(_lib.lib().fgt_conv2d(*args, _stream()), "fgt_conv2d")
            return {0: out, 1: out_s, 2: (out, out_s)}[osp]
    check(_lib.lib().fgt_conv2d(*args, _stream()), "fgt_conv2d")
    return {0: out, 1: out_s, 2: (out, out_s)}[osp]


def _aliases(out, out_s, *ins):
    """True when an output buffer overlaps one of the input / aux tensors (in-place epilogues).  Channel slices of one wide
    channels-last buffer (same row stride, disjoint channel windows: the concat buffers of RAFT / the token buffers) do not overlap."""
    def box(t):
        if t is None:
            return None
        t = t.data if isinstance(t, Split) else t
        if t.numel() == 0:
            return None
        lo = t.storage_offset()
        hi = lo + sum((n - 1) * st for n, st in zip(t.shape, t.stride())) + 1             # element range inside the storage
        outer = [st for n, st in zip(t.shape[:-1], t.stride()[:-1]) if n > 1 and st > 0]
        ld = min(outer, default=0)
        # the channel-window shortcut below is only valid when EVERY outer stride is a multiple of the pixel stride (padded rows /
        # frames break it): otherwise the tensor counts as overlapping whatever shares its element range (conservative)
        regular = bool(ld) and all(st % ld == 0 for st in outer)
        return (t.untyped_storage().data_ptr(), t.element_size(), lo, hi, ld, t.shape[-1] if t.dim() else 1, regular)

    def overlap(a, b):
        if a[0] != b[0] or a[1] != b[1]:
            # different storages (or element sizes): compare byte ranges
            a0, a1 = a[0] + a[2] * a[1], a[0] + a[3] * a[1]
            b0, b1 = b[0] + b[2] * b[1], b[0] + b[3] * b[1]
            return a0 < b1 and b0 < a1
        if not (a[2] < b[3] and b[2] < a[3]):
            return False
        if a[4] and a[4] == b[4] and a[5] <= a[4] and b[5] <= b[4] and a[6] and b[6]:     # same pixel stride: compare channel windows
            ca, cb = a[2] % a[4], b[2] % b[4]
            if ca + a[5] <= a[4] and cb + b[5] <= b[4]:
                return ca < cb + b[5] and cb < ca + a[5]
        return True

    outs = [o for o in (box(out), box(out_s)) if o]
    return any(overlap(o, i) for o in outs for i in (box(t) for t in ins) if i)


def _autotune(d, args, candidates=None):
    """Time each tile candidate on the current stream (1 warm + 4 timed launches; candidates of a family differ by 2-10 %, two launches
    were within the noise) and return the fastest tile code."""
    fn, st = _lib.lib().fgt_conv2d, _stream()
    best, best_ms = 0, None
    for name in (candidates or TILE_CANDIDATES):
        d.tile = TILE[name]
        if fn(*args, st) != 0:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn(*args, st)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        if best_ms is None or ms < best_ms:
            best, best_ms = TILE[name], ms
    d.tile = 0
    return best


_bgemm_tiles = {}
BGEMM_CANDIDATES = ["128x128w", "128x128eaw", "128x128x8w", "128x128x8eaw", "256x128w", "256x128eaw", "256x128x16w", "256x128x16eaw"]


def batched_gemm_nt(a, b, out, scale=1.0):
    """out[g] = scale * a[g] @ b[g]^T for interleaved Splits a [G, M, K], b [G, N, K] and an fp32 out [G, M, N]: ONE fgt_conv2d launch in the batched
    GEMM mode of ABI 8 (fgt_conv_desc.gb_*; csrc/conv_wide.hip) — b is used as it lies, as the weight image of its group.  bf16x3 products, fp32
    accumulation over K in 32-channel steps.  RAFT's all-pairs correlation volume (RAFT/corr.py:52-60) for a whole pair batch."""
    assert isinstance(a, Split) and isinstance(b, Split) and a.il and b.il, "batched_gemm_nt: interleaved Splits"
    G, M, K = a.shape
    Gb, N, Kb = b.shape
    assert (Gb, Kb) == (G, K) and tuple(out.shape) == (G, M, N) and out.is_contiguous() and out.dtype == torch.float32
    assert a.data.is_contiguous() and b.data.is_contiguous() and K % 32 == 0 and N % 8 == 0
    _require_dev(out)
    d = ConvDesc()
    d.N, d.H, d.W = 1, 1, M
    d.C0, d.ld0, d.off0 = G * K, 2 * K, 0
    d.Cout, d.groups = G * N, G
    d.kh = d.kw = d.sh = d.sw = d.dh = d.dw = 1
    d.Ho, d.Wo, d.ldo = 1, M, N
    d.slope, d.out_scale = 0.2, float(scale)
    d.Kpad, d.Npad = K, ceil_to(N, 128)
    d.precision, d.in_split, d.w_il = PREC["bf16x3"], 2, 1
    d.gb_x0, d.gb_w, d.gb_o = M * 2 * K, N * 2 * K, M * N
    args = (C.byref(d), _ptr(a.data), _ptr(None), _ptr(b.data), _ptr(None), _ptr(None), _ptr(None), _ptr(None), _ptr(out), _ptr(None))
    key = (G, M, N, K)
    best = _bgemm_tiles.get(key)
    if best is None:
        if torch.cuda.is_current_stream_capturing():
            best = TILE["128x128eaw"]
        else:
            best = _bgemm_tiles[key] = _autotune(d, args, BGEMM_CANDIDATES) or TILE["128x128w"]
    d.tile = best
    check(_lib.lib().fgt_conv2d(*args, _stream()), "fgt_conv2d (batched GEMM)")
    return out


def linear(x, pc, **kw):
    """Linear on a [rows, C] token matrix (a 1x1 conv over a 1 x rows image).  x / x1 may be Splits; out_split as conv2d."""
    rows = x.shape[0]
    out = kw.pop("out", None)
    osp = kw.get("out_split")
    if out is None and osp != "only":
        out = torch.empty(rows, pc.Cout, dtype=torch.float32, device=x.device)
    out_s = kw.pop("out_s", None)
    if osp and out_s is None:
        out_s = Split.empty((rows, pc.Cout), x.device)
    as_img = lambda t: None if t is None else (t.view(1, 1, *t.shape) if isinstance(t, Split) else t.unsqueeze(0).unsqueeze(0))
    conv2d(as_img(x), pc, x1=as_img(kw.pop("x1", None)), out=as_img(out), out_s=as_img(out_s), **kw)
    return {None: out, False: out, "only": out_s, "both": (out, out_s)}[osp]


def _out_desc(t, allow_il=False):
    """(pointer-holding tensor, row stride, plane stride) of an fp32 tensor (ps = 0) or a Split (interleaved: ps = 32, row stride in
    elements of the 2C-wide rows — fgt_layernorm / fgt_fold; the attention kernels write planes or fp16 only)."""
    if isinstance(t, Split):
        assert allow_il or not t.il, "this producer writes the planes layout (or fp16)"
        return t.hi, t.hi.stride(0), t.ps
    return t, (0 if t is None else t.stride(0)), 0


def layernorm(x0, gA, bA, x1=None, gB=None, bB=None, outA=None, outB=None, eps=1e-5, splitA=False, splitB=False):
    """Row LayerNorm over [x0 | x1]; optional second affine output sharing the statistics.  splitA / splitB (or passing a
    Split as outA / outB) writes that output pre-split for the GEMM that consumes it (interleaved when split_il() says so)."""
    _require_dev(x0, x1, gA, bA, gB, bB)
    rows, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    if outA is None:
        outA = Split.empty((rows, C0 + C1), x0.device, split_il(C0 + C1)) if splitA else torch.empty(rows, C0 + C1, dtype=torch.float32, device=x0.device)
    if gB is not None and outB is None:
        outB = Split.empty((rows, C0 + C1), x0.device, split_il(C0 + C1)) if splitB else torch.empty(rows, C0 + C1, dtype=torch.float32, device=x0.device)
    (tA, ldA, psA), (tB, ldB, psB) = _out_desc(outA, True), _out_desc(outB, True)
    check(_lib.lib().fgt_layernorm(_ptr(x0), C0, x0.stride(0), _ptr(x1), C1, 0 if x1 is None else x1.stride(0), rows, eps,
                                   _ptr(gA), _ptr(bA), _ptr(tA), ldA, _ptr(gB), _ptr(bB), _ptr(tB), ldB, psA, psB, _stream()),
          "fgt_layernorm")
    return (outA, outB) if gB is not None else outA


def _attn_prec(insp, h16, precision):
    """Arithmetic of an attention call: the input format decides for Splits; fp32 inputs in the 'f16' mode run in bf16x3."""
    if insp:
        return PREC["f16" if h16 else "bf16x3"]
    prec = precision if precision is not None else DEFAULT_ATTN_PRECISION
    return PREC["bf16x3" if prec == "f16" else prec]


def _attn_out(rows, c, device, out_split, d, h16=False):
    # an fp16 output needs fp16 inputs (csrc/attention_split.hip); with other inputs the split output is the bf16 pair
    out = Split.empty((rows, c), device, h=bool(h16)) if out_split else torch.empty(rows, c, dtype=torch.float32, device=device)
    t, ld, ps = _out_desc(out)
    d.ldo, d.out_split, d.pso = ld, int(bool(out_split)), ps
    return out, t


def _attn_in(t):
    """(tensor holding the data pointer, row stride, plane stride) of an fp32 tensor or a planes-layout Split input."""
    if isinstance(t, Split):
        assert not t.il, "attention takes the planes layout (or fp16)"
        return t.hi, t.hi.stride(0), t.ps
    return t, (0 if t is None else t.stride(0)), 0


def attention_temporal(qkv, b, t, nh, nw, heads, group, c, precision=None, out_split=False, tq=None):
    """Temporal zone attention reading q/k/v in place from a fused [b*t*nh*nw, 3c] projection buffer (fp32, or a Split written by the
    QKV GEMM: K / V tiles then stream through LDS-DMA, csrc/attention_split.hip).
    tq: only the first tq frames of every batch element are queried (keys / values: all t frames); the result has b*tq*nh*nw rows."""
    insp = isinstance(qkv, Split)
    if not insp:
        _require_dev(qkv)
    d = AttnDesc()
    d.mode, d.b, d.t, d.h, d.w, d.nh, d.nw, d.heads, d.group = 0, b, t, nh, nw, nh, nw, heads, group
    d.ws, d.n_global = 0, 0
    src, ld, ps = _attn_in(qkv)
    d.ldq = d.ldk = d.ldv = ld
    d.qoff, d.koff, d.voff = 0, c, 2 * c
    d.ldg_k = d.ldg_v = 0
    h16 = insp and qkv.h
    d.in_split, d.psq, d.psk, d.psv = (2 if h16 else int(insp)), ps, ps, ps
    tq = t if tq is None else int(tq)
    assert 0 < tq <= t
    d.tq = 0 if tq == t else tq
    out, optr = _attn_out(b * tq * nh * nw, c, src.device, out_split, d, h16)
    d.precision = _attn_prec(insp, h16, precision)
    check(_lib.lib().fgt_attention(C.byref(d), _ptr(src), _ptr(src), _ptr(src), None, None, _ptr(optr), _stream()),
          "fgt_attention(temporal)")
    return out


def attention_spatial(q, k, v, kg, vg, bt, h, w, nh, nw, heads, ws, n_global, precision=None, out_split=False, pad_row=None):
    """Window attention + shared global tokens; q/k/v are [bt*nh*nw, c] maps on the padded grid, output cropped.  All five inputs fp32, or
    all five Splits (written by the projection GEMMs).
    pad_row: the maps are COMPACT — rows [0, bt*h*w) are the real tokens and every zero-padded position of the window grid reads row
    `pad_row` of q / k / v (the projections of a padded token are one constant row, so the padded rows are never computed)."""
    insp = isinstance(q, Split)
    assert all(isinstance(x, Split) == insp for x in (k, v, kg, vg)), "attention_spatial: inputs must all be fp32 or all be Splits"
    if not insp:
        _require_dev(q, k, v, kg, vg)
    c = q.shape[1]
    d = AttnDesc()
    d.mode, d.b, d.t, d.h, d.w, d.nh, d.nw, d.heads, d.group = 1, 1, bt, h, w, nh, nw, heads, 0
    d.ws, d.n_global = ws, n_global
    (qt, d.ldq, d.psq), (kt, d.ldk, d.psk), (vt, d.ldv, d.psv) = _attn_in(q), _attn_in(k), _attn_in(v)
    (kgt, d.ldg_k, d.psg_k), (vgt, d.ldg_v, d.psg_v) = _attn_in(kg), _attn_in(vg)
    d.qoff = d.koff = d.voff = 0
    h16 = insp and q.h
    assert not insp or all(x.h == h16 for x in (k, v, kg, vg)), "attention_spatial: one Split format for all inputs"
    d.in_split = 2 if h16 else int(insp)
    if pad_row is not None:
        assert 0 <= pad_row < min(x.shape[0] for x in (q, k, v))
        d.compact, d.pad_row = 1, int(pad_row)
    out, optr = _attn_out(bt * h * w, c, qt.device, out_split, d, h16)
    d.precision = _attn_prec(insp, h16, precision)
    check(_lib.lib().fgt_attention(C.byref(d), _ptr(qt), _ptr(kt), _ptr(vt), _ptr(kgt), _ptr(vgt), _ptr(optr), _stream()),
          "fgt_attention(spatial)")
    return out


def dw_pool(x0, x1, bt, nh, nw, k, w, bias, out, h=None, w_real=None):
    """Depthwise kxk/stride-k conv over the [x0 | x1] maps on the nh x nw window grid -> out [bt*(nh/k)*(nw/k), C0+C1].
    (h, w_real): the maps hold only the h x w_real real tokens of every frame ([bt*h*w_real, C]); the rest of the grid is zero padding."""
    _require_dev(x0, x1, w, bias, out)
    vh, vw = (nh if h is None else h), (nw if w_real is None else w_real)
    check(_lib.lib().fgt_dw_pool(_ptr(x0), x0.shape[1], x0.stride(0), _ptr(x1), 0 if x1 is None else x1.shape[1],
                                 0 if x1 is None else x1.stride(0), bt, nh, nw, vh, vw, k, _ptr(w), _ptr(bias), _ptr(out),
                                 out.stride(0), _stream()), "fgt_dw_pool")
    return out


def dw3x3_residual(x, bt, h, w, wgt, bias):
    _require_dev(x, wgt, bias)
    assert x.is_contiguous()
    out = torch.empty_like(x)
    check(_lib.lib().fgt_dw3x3_residual(_ptr(x), bt, h, w, x.shape[-1], _ptr(wgt), _ptr(bias), _ptr(out), _stream()),
          "fgt_dw3x3_residual")
    return out


def fold(Y, frames, th, tw, Cc, k, s, p, Hf, Wf, normalize, res=None, out=None, relu=False, out_split=False):
    """Overlap-add of [frames*th*tw, k*k*Cc] (tap-major columns) to [frames, Hf, Wf, Cc]; relu: max(., 0) last;
    out_split: write the result pre-split (the FFN's second Linear consumes it as a conv over this map).
    Y: fp32, or the fp16 Split a GEMM wrote in the 'f16' mode (half the bytes of the block's largest tensor; fp32 sums)."""
    y_h = isinstance(Y, Split)
    if y_h:
        assert Y.h, "fold: a Split input must be the fp16 format"
        Y = Y.data
    else:
        _require_dev(Y)
    _require_dev(res)
    if out is None:
        out = Split.empty((frames, Hf, Wf, Cc), Y.device, split_il(Cc)) if out_split else torch.empty(frames, Hf, Wf, Cc, dtype=torch.float32, device=Y.device)
    ldres = 0 if res is None else _as_map(res)[5]
    if isinstance(out, Split):
        optr, ldo, ps = out.hi, _as_map(out.hi)[5], out.ps
    else:
        _require_dev(out)
        optr, ldo, ps = out, _as_map(out)[5], 0
    check(_lib.lib().fgt_fold(_ptr(Y), Y.stride(0), frames, th, tw, Cc, k, s, p, Hf, Wf, int(normalize), _ptr(res), ldres,
                              _ptr(optr), ldo, int(relu), ps, int(y_h), _stream()), "fgt_fold")
    return out


def nchw_to_nhwc(src, dst, coff=0, zero_to=0, scale=1.0, shift=0.0):
    """src [N,C,H,W] contiguous -> dst[..., coff:coff+C] of a channels-last buffer [N,H,W,ld]."""
    _require_dev(src, dst)
    src = src.contiguous()
    N, Cc, H, W = src.shape
    _, dN, dH, dW, dC, ldd = _as_map(dst)
    assert (dN, dH, dW) == (N, H, W)
    check(_lib.lib().fgt_nchw_to_nhwc(_ptr(src), N, Cc, H, W, _ptr(dst), ldd, coff, zero_to, scale, shift, _stream()),
          "fgt_nchw_to_nhwc")
    return dst


def nhwc_to_nchw(src):
    _require_dev(src)
    src4, N, H, W, Cc, lds = _as_map(src)
    out = torch.empty(N, Cc, H, W, dtype=torch.float32, device=src.device)
    check(_lib.lib().fgt_nhwc_to_nchw(_ptr(src4), lds, 0, N, Cc, H, W, _ptr(out), _stream()), "fgt_nhwc_to_nchw")
    return out


def pad_tokens(src, bt, h, w, nh, nw, out=None):
    """[bt*h*w, C] -> [bt*nh*nw, C]: zero pad where (nh,nw) is larger, crop where it is smaller."""
    _require_dev(src, out)
    Cc = src.shape[1]
    if out is None:
        out = torch.empty(bt * nh * nw, Cc, dtype=torch.float32, device=src.device)
    check(_lib.lib().fgt_pad_tokens(_ptr(src), src.stride(0), bt, h, w, Cc, nh, nw, _ptr(out), out.stride(0), _stream()),
          "fgt_pad_tokens")
    return out


def warp(img, flow, align_corners=False, absolute=False):
    """img [B,H,W,C] channels-last, flow [B,H,W,2] -> bilinear backward warp (zeros padding)."""
    _require_dev(img, flow)
    img4, B, H, W, Cc, ldi = _as_map(img)
    flow = flow.contiguous()
    out = torch.empty(B, H, W, Cc, dtype=torch.float32, device=img.device)
    check(_lib.lib().fgt_warp(_ptr(img4), ldi, _ptr(flow), B, H, W, Cc, int(align_corners), int(absolute), _ptr(out), Cc,
                              _stream()), "fgt_warp")
    return out


def fb_consistency(flow_fw, flow_bw, alpha1=0.01, alpha2=0.5):
    """flows [B,H,W,2] channels-last -> (occ_fw, occ_bw) [B,H,W]."""
    _require_dev(flow_fw, flow_bw)
    flow_fw, flow_bw = flow_fw.contiguous(), flow_bw.contiguous()
    B, H, W, _ = flow_fw.shape
    o1 = torch.empty(B, H, W, dtype=torch.float32, device=flow_fw.device)
    o2 = torch.empty_like(o1)
    check(_lib.lib().fgt_fb_consistency(_ptr(flow_fw), _ptr(flow_bw), B, H, W, alpha1, alpha2, _ptr(o1), _ptr(o2), _stream()),
          "fgt_fb_consistency")
    return o1, o2


def avgpool2(src, rows, H, W):
    _require_dev(src)
    out = torch.empty(rows, H // 2, W // 2, dtype=torch.float32, device=src.device)
    check(_lib.lib().fgt_avgpool2(_ptr(src), rows, H, W, _ptr(out), _stream()), "fgt_avgpool2")
    return out


def corr_lookup(pyr, B, H1, W1, radius, coords, out=None, out_s=None):
    """pyr: list of level volumes; coords [B,H1,W1,2]; out channels-last [B,H1,W1,levels*(2r+1)^2] (may be a slice) and / or out_s, a planes
    `Split` of >= that many channels (a multiple of 32 for the LDS-DMA conv kernels: the extra channels are written as zeros)."""
    _require_dev(coords, out, *pyr)
    arr = (C.c_void_p * len(pyr))(*[p.data_ptr() for p in pyr])
    ldo = 0 if out is None else _as_map(out)[5]
    ld_s = ps = nch_pad = 0
    if out_s is not None:
        assert not out_s.il and not out_s.h, "corr_lookup writes the planes layout"
        _, sN, sH, sW, nch_pad, ld_s = _as_map(out_s.hi)
        assert sN * sH * sW == B * H1 * W1
        ps = out_s.ps
    check(_lib.lib().fgt_corr_lookup_split(arr, len(pyr), B, H1, W1, radius, _ptr(coords.contiguous()), _ptr(out), ldo,
                                           _ptr(None if out_s is None else out_s.data), ld_s, ps, nch_pad, _stream()), "fgt_corr_lookup")
    return out if out_s is None else (out_s if out is None else (out, out_s))


def convex_upsample(flow, mask):
    """flow [B,H,W,2(+pad)] channels-last view, mask [B,H,W,576] -> [B,2,8H,8W] NCHW."""
    _require_dev(flow, mask)
    f4, B, H, W, _, ldf = _as_map(flow)
    m4, _, _, _, _, ldm = _as_map(mask)
    out = torch.empty(B, 2, 8 * H, 8 * W, dtype=torch.float32, device=flow.device)
    check(_lib.lib().fgt_convex_upsample(_ptr(f4), ldf, _ptr(m4), ldm, B, H, W, _ptr(out), _stream()), "fgt_convex_upsample")
    return out


def instnorm(x, act=None, res=None, act2=None, eps=1e-5, out=None, out_split=None, out_s=None):
    """InstanceNorm2d(affine=False) over a channels-last map, fused act / residual / act2.  out_split: None -> fp32; "only" -> a Split (what the next conv's
    LDS-DMA loader reads); "both" -> (fp32, Split)."""
    _require_dev(x, res, out)
    x4, N, H, W, Cc, ld = _as_map(x)
    stats = torch.empty(N * Cc * 2, dtype=torch.float64, device=x.device)
    check(_lib.lib().fgt_instnorm_stats(_ptr(x4), ld, N, H * W, Cc, _ptr(stats), _stream()), "fgt_instnorm_stats")
    if out is None and out_split != "only":
        out = torch.empty(N, H, W, Cc, dtype=torch.float32, device=x.device)
    if out_split and out_s is None:
        out_s = Split.empty((N, H, W, Cc), x.device, interleaved=split_il(Cc), h=False)
    ldres = 0 if res is None else _as_map(res)[5]
    lds_, ps_ = (0, 0) if out_s is None else (_as_map(out_s.hi)[5], out_s.ps)
    check(_lib.lib().fgt_instnorm_apply_split(_ptr(x4), ld, N, H * W, Cc, _ptr(stats), eps, ACT[act], _ptr(res), ldres, ACT[act2],
                                              _ptr(out), 0 if out is None else _as_map(out)[5], _ptr(None if out_s is None else out_s.data), lds_, ps_, _stream()),
          "fgt_instnorm_apply")
    return {None: out, False: out, "only": out_s, "both": (out, out_s)}[out_split]


def axpby(a, sa=1.0, b=None, sb=1.0, act=None, out=None, slope=0.2):
    """out = act(a*sa + b*sb) over [rows, C] views."""
    _require_dev(a, b, out)
    a2 = a.reshape(-1, a.shape[-1]) if a.is_contiguous() else a
    rows, Cc = a2.shape
    if out is None:
        out = torch.empty(rows, Cc, dtype=torch.float32, device=a.device)
    check(_lib.lib().fgt_axpby(_ptr(a2), a2.stride(0), sa, _ptr(b), 0 if b is None else b.stride(0), sb, rows, Cc, ACT[act],
                               slope, _ptr(out), out.stride(0), _stream()), "fgt_axpby")
    return out


def pack_frames(frames01, masks, ids=None, out=None):
    """fgt_pack_frames: frames01 [N,3,H,W] in [0,1], masks [N,1,H,W] -> [n,H,W,4] = ((f*2-1)*(1-m) | m) for frames ids (int32) or all."""
    _require_dev(frames01, masks, out)
    assert frames01.is_contiguous() and masks.is_contiguous() and frames01.shape[1] == 3 and masks.shape[1] == 1
    N, _, H, W = frames01.shape
    n = N if ids is None else ids.numel()
    assert ids is None or (ids.dtype == torch.int32 and ids.is_cuda)
    if out is None:
        out = torch.empty(n, H, W, 4, dtype=torch.float32, device=frames01.device)
    o4, oN, oH, oW, oC, ldo = _as_map(out)
    assert (oN, oH, oW) == (n, H, W) and oC >= 4
    check(_lib.lib().fgt_pack_frames(_ptr(frames01), _ptr(masks), C.c_void_p(0 if ids is None else ids.data_ptr()), n, H, W, _ptr(o4), ldo,
                                     _stream()), "fgt_pack_frames")
    return out


def norm_flows(flows, n_out=None):
    """fgt_norm_flows on [..., n, C, H, W] (leading dims of size 1 allowed): every (frame, channel) map divided by its signed
    maximum (tool/video_inpainting.py:402-407); n_out = n + 1 also duplicates the last flow (:705)."""
    _require_dev(flows)
    lead = flows.shape[:-4]
    assert all(d == 1 for d in lead), "norm_flows: one clip at a time"
    f = flows.reshape(flows.shape[-4:]).contiguous()
    n, Cc, H, W = f.shape
    n_out = n if n_out is None else n_out
    out = torch.empty((n_out, Cc, H, W), dtype=torch.float32, device=f.device)
    check(_lib.lib().fgt_norm_flows(_ptr(f), n, n_out, Cc, H * W, _ptr(out), _stream()), "fgt_norm_flows")
    return out.view(*lead, n_out, Cc, H, W)


def gather_rows(src, ids, out=None):
    """out[i] = src[ids[i]] over the leading dimension (ids: int32 device tensor); src[j] must be contiguous."""
    _require_dev(src, out)
    assert ids.dtype == torch.int32 and ids.is_cuda and src[0].is_contiguous()
    n, row = ids.numel(), src[0].numel()
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    assert out.shape[0] == n and out[0].is_contiguous() and out[0].numel() == row
    check(_lib.lib().fgt_gather_rows(_ptr(src), src.stride(0), C.c_void_p(ids.data_ptr()), n, row, _ptr(out), out.stride(0), _stream()),
          "fgt_gather_rows")
    return out


def compose_blend(out_nchw, ids, first, frames01, masks, comp):
    _require_dev(out_nchw, frames01, masks, comp)
    n = ids.numel()
    H, W = out_nchw.shape[-2:]
    check(_lib.lib().fgt_compose_blend(_ptr(out_nchw.contiguous()), C.c_void_p(ids.data_ptr()), C.c_void_p(first.data_ptr()), n,
                                       _ptr(frames01), _ptr(masks), H, W, _ptr(comp), _stream()), "fgt_compose_blend")
    return comp


def quantize_u8(x, out=None):
    """fgt_quantize_u8: fp32 values in (-1,1) -> uint8 astype(uint8)((x+1)/2*255) (tool/video_inpainting.py:725-726,731)."""
    _require_dev(x)
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() == x.numel()
    check(_lib.lib().fgt_quantize_u8(_ptr(x), x.numel(), C.c_void_p(out.data_ptr()), _stream()), "fgt_quantize_u8")
    return out


def compose_blend_u8(filled_u8, ids, first, frames01, masks, comp):
    _require_dev(frames01, masks, comp)
    assert filled_u8.dtype == torch.uint8 and filled_u8.is_cuda and filled_u8.is_contiguous()
    n = ids.numel()
    H, W = filled_u8.shape[-2:]
    check(_lib.lib().fgt_compose_blend_u8(C.c_void_p(filled_u8.data_ptr()), C.c_void_p(ids.data_ptr()), C.c_void_p(first.data_ptr()), n,
                                          _ptr(frames01), _ptr(masks), H, W, _ptr(comp), _stream()), "fgt_compose_blend_u8")
    return comp


_prof_on = False


# which implementation the two sparse solves take: 'auto' = one workgroup per problem on chip when every hole's bounding box fits
# (csrc/solve_onchip.hip), else the multi-launch kernels; 'onchip' / 'multilaunch' force one (tests, A/B measurements)
SOLVER = os.environ.get("FGT_SOLVER", "auto")
last_solver = {}          # what the last laplace_fill / poisson_blend call used (bench accounting, tests)


def mask_bbox(m8):
    """fgt_mask_bbox: uint8 masks [n, H, W] -> device int32 [n, 4] = (y0, x0, y1, x1) inclusive per mask (empty: y1 < y0)."""
    assert m8.dtype == torch.uint8 and m8.is_cuda and m8.is_contiguous() and m8.dim() == 3
    bb = torch.empty(m8.shape[0], 4, dtype=torch.int32, device=m8.device)
    check(_lib.lib().fgt_mask_bbox(C.c_void_p(m8.data_ptr()), m8.shape[0], m8.shape[1], m8.shape[2], C.c_void_p(bb.data_ptr()), _stream()), "fgt_mask_bbox")
    return bb


def _onchip_bounds(bb):
    """(max rows, max cols) of the boxes — the ONE host read-back of the on-chip solvers (they size their workgroup and LDS by it) — or None
    when such a box cannot fit one workgroup: <= 6144 strips of 4 cells (~24 k cells), (rows + 2) * (strip columns + 8) floats + scratch <= 160 KB."""
    b = bb.cpu()
    ok = b[:, 2] >= b[:, 0]
    if not bool(ok.any()):
        return 0, 0
    rows = int((b[ok, 2] - b[ok, 0] + 1).max())
    cols = int((b[ok, 3] - b[ok, 1] + 1).max())
    wq = (cols + 6) // 4                   # strips start at a multiple of 4 columns: up to 3 extra cells per row
    if rows * wq > 512 * 12 or ((rows + 2) * (wq * 4 + 8)) * 4 + 256 > 160 * 1024:
        return None
    return rows, cols


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def hole_bounds(masks):
    """(max rows, max cols) of the holes' bounding boxes of uint8 / bool masks [n, H, W], or None when a box does not fit one workgroup.
    ONE blocking read-back: a clip's masks are fixed, so callers compute this once per clip and pass it as `bounds=` to laplace_fill /
    poisson_blend (both directions of the diffusion fill and the Poisson blend use the same holes) instead of paying a device-to-host
    sync per call."""
    m8 = (masks != 0).to(torch.uint8).contiguous()
    return _onchip_bounds(mask_bbox(m8))


def solver_report(name):
    """What the last `laplace_fill` / `poisson_blend` call did, read back from the device: solver, problems, iterations (mean / max), the
    number of problems that hit the iteration cap and the number the device NaN-filled because their box exceeded the promised bounds
    (status 1; cannot happen with bounds from `hole_bounds` of the same masks).  FGT_SOLVER_CHECK=1 makes the two entry points raise
    RuntimeError after a call in which either count is non-zero (a blocking read-back: debugging only)."""
    st = last_solver.get(name, {})
    if st.get("solver") != "onchip":
        return {"solver": st.get("solver", "none")}
    s = st["status"].cpu().long()
    it = s >> 1
    return {"solver": "onchip", "problems": int(s.numel()), "iterations_mean": float(it.float().mean()), "iterations_max": int(it.max()),
            "cap_hits": int((it >= st.get("iters", 1 << 30)).sum()), "nan_filled": int((s == 1).sum()), "bbox": st.get("bbox")}


def _solver_check(name):
    if os.environ.get("FGT_SOLVER_CHECK") == "1" and not _capturing():
        rep = solver_report(name)
        if rep.get("nan_filled", 0):
            raise RuntimeError(f"{name}: {rep['nan_filled']} problems exceeded the promised bounding box (NaN-filled): `bounds` came from other masks?")
        if rep.get("cap_hits", 0):
            raise RuntimeError(f"{name}: {rep['cap_hits']} of {rep['problems']} problems hit the iteration cap without reaching the tolerance")


def laplace_fill(maps, masks, iters=1000, tol=1e-6, solver=None, bounds=None):
    """fgt_laplace_fill[_onchip]: maps [B, H, W] fp32, masks [n_masks, H, W] (non-zero = hole; map b uses mask b % n_masks) -> filled [B, H, W].
    tool/utils/region_fill.py:7-63 for every map at once by conjugate gradients (`iters` = iteration cap, a map stops at tol * |r0|).
    `bounds` = hole_bounds(masks) computed earlier (no host sync in this call, legal under stream capture); without it the on-chip path
    reads the boxes back once (skipped while a stream is capturing: the multi-launch kernels run then)."""
    _require_dev(maps)
    assert maps.dim() == 3 and masks.dim() == 3 and masks.shape[1:] == maps.shape[1:] and masks.is_cuda
    maps = maps.contiguous()
    m8 = (masks != 0).to(torch.uint8).contiguous()
    B, H, W = maps.shape
    out = torch.empty_like(maps)
    solver = solver or SOLVER
    if solver != "multilaunch" and W % 4 == 0 and (bounds is not None or not _capturing()):
        bb = mask_bbox(m8)
        if bounds is None:
            bounds = _onchip_bounds(bb)
        if bounds is not None:
            status = torch.empty(B, dtype=torch.int32, device=maps.device)
            check(_lib.lib().fgt_laplace_fill_onchip(_ptr(maps), C.c_void_p(m8.data_ptr()), C.c_void_p(bb.data_ptr()), B, m8.shape[0], H, W, _ptr(out),
                                                     bounds[0], bounds[1], int(iters), float(tol), C.c_void_p(status.data_ptr()), _stream()), "fgt_laplace_fill_onchip")
            last_solver["laplace_fill"] = {"solver": "onchip", "status": status, "bbox": bounds, "iters": int(iters)}
            _solver_check("laplace_fill")
            return out
        if solver == "onchip":
            raise RuntimeError("laplace_fill(solver='onchip'): a hole's bounding box does not fit one workgroup")
    last_solver["laplace_fill"] = {"solver": "multilaunch"}
    ws = torch.empty(_lib.lib().fgt_laplace_fill_workspace(B, H, W), dtype=torch.uint8, device=maps.device)
    check(_lib.lib().fgt_laplace_fill(_ptr(maps), _ptr(m8), B, m8.shape[0], H, W, _ptr(out), _ptr(ws), int(iters), float(tol), _stream()),
          "fgt_laplace_fill")
    return out


def flow_propagate(gx, gy, mask, flow_f, flow_b, consistency_thres=5.0, alpha=0.1, tab=32):
    """fgt_flow_propagate: gx, gy [N,H,W,3] fp32, mask [N,H,W] (non-zero = hole), flow_f / flow_b [N-1,H,W,2] fp32 ->
    (gx, gy [N,H,W,3], mask_tofill [N,H,W] bool).  tool/get_flowNN_gradient.py:11-534 for the whole clip in one call."""
    _require_dev(gx, gy, flow_f, flow_b)
    N, H, W, _ = gx.shape
    assert gy.shape == gx.shape and gx.shape[-1] == 3 and tuple(mask.shape) == (N, H, W) and mask.is_cuda
    assert N == 1 or (tuple(flow_f.shape) == (N - 1, H, W, 2) and tuple(flow_b.shape) == (N - 1, H, W, 2))
    gx, gy = gx.contiguous(), gy.contiguous()
    m8 = (mask != 0).to(torch.uint8).contiguous()
    ff = None if N == 1 else flow_f.contiguous()
    fb = None if N == 1 else flow_b.contiguous()
    ox, oy = torch.empty_like(gx), torch.empty_like(gy)
    fill = torch.empty(N, H, W, dtype=torch.uint8, device=gx.device)
    ws = torch.empty(_lib.lib().fgt_flow_propagate_workspace(N, H, W), dtype=torch.uint8, device=gx.device)
    check(_lib.lib().fgt_flow_propagate(_ptr(gx), _ptr(gy), C.c_void_p(m8.data_ptr()), _ptr(ff), _ptr(fb), N, H, W, float(consistency_thres),
                                        float(alpha), int(tab), _ptr(ox), _ptr(oy), C.c_void_p(fill.data_ptr()), C.c_void_p(ws.data_ptr()), _stream()),
          "fgt_flow_propagate")
    return ox, oy, fill.bool()


def poisson_blend(target, gx, gy, hole, gmask, iters=2000, tol=1e-7, solver=None, bounds=None):
    """fgt_poisson_blend[_onchip]: target, gx, gy [N,H,W,3] fp32; hole, gmask [N,H,W] (non-zero = hole / gradient unknown) ->
    (blend [N,H,W,3], unfilled [N,H,W] bool).  tool/utils/Poisson_blend_img.py:19-244 for the whole clip."""
    _require_dev(target, gx, gy)
    N, H, W, _ = target.shape
    assert gx.shape == target.shape and gy.shape == target.shape and tuple(hole.shape) == (N, H, W) and tuple(gmask.shape) == (N, H, W)
    target, gx, gy = target.contiguous(), gx.contiguous(), gy.contiguous()
    h8, g8 = (hole != 0).to(torch.uint8).contiguous(), (gmask != 0).to(torch.uint8).contiguous()
    out = torch.empty_like(target)
    unf = torch.empty(N, H, W, dtype=torch.uint8, device=target.device)
    ws = torch.empty(_lib.lib().fgt_poisson_blend_workspace(N, H, W), dtype=torch.uint8, device=target.device)
    solver = solver or SOLVER
    if solver != "multilaunch" and W % 4 == 0 and (bounds is not None or not _capturing()):
        bb = mask_bbox(h8)
        if bounds is None:
            bounds = _onchip_bounds(bb)
        if bounds is not None:
            status = torch.empty(3 * N, dtype=torch.int32, device=target.device)
            check(_lib.lib().fgt_poisson_blend_onchip(_ptr(target), _ptr(gx), _ptr(gy), C.c_void_p(h8.data_ptr()), C.c_void_p(g8.data_ptr()), C.c_void_p(bb.data_ptr()),
                                                      N, H, W, bounds[0], bounds[1], int(iters), float(tol), _ptr(out), C.c_void_p(unf.data_ptr()),
                                                      C.c_void_p(status.data_ptr()), C.c_void_p(ws.data_ptr()), _stream()), "fgt_poisson_blend_onchip")
            last_solver["poisson_blend"] = {"solver": "onchip", "status": status, "bbox": bounds, "iters": int(iters)}
            _solver_check("poisson_blend")
            return out, unf.bool()
        if solver == "onchip":
            raise RuntimeError("poisson_blend(solver='onchip'): a hole's bounding box does not fit one workgroup")
    last_solver["poisson_blend"] = {"solver": "multilaunch"}
    check(_lib.lib().fgt_poisson_blend(_ptr(target), _ptr(gx), _ptr(gy), C.c_void_p(h8.data_ptr()), C.c_void_p(g8.data_ptr()), N, H, W,
                                       int(iters), float(tol), _ptr(out), C.c_void_p(unf.data_ptr()), C.c_void_p(ws.data_ptr()), _stream()),
          "fgt_poisson_blend")
    return out, unf.bool()


def prof_enable(on, kinds=None):
    """Per-launch HIP-event timing on / off; kinds: iterable of PROF_KINDS names to restrict it to (None = every kind)."""
    global _prof_on
    _prof_on = bool(on)
    if on and kinds is not None:
        mask = 0
        for k in kinds:
            mask |= 1 << PROF_KINDS[k]
        _lib.lib().fgt_prof_enable_kinds(mask)
    else:
        _lib.lib().fgt_prof_enable(int(on))


def prof_is_enabled():
    return _prof_on


PROF_KINDS = {"conv": 0, "attn_temporal": 1, "attn_spatial": 2, "layernorm": 3, "fold": 4, "conv_small": 5, "dw_pool": 6, "warp": 7,
              "corr_lookup": 8, "pointwise": 9, "all": -1}


def prof_collect(kind="conv"):
    """(total ms, total algorithmic flops, launches, total unique bytes) of the launches of `kind` recorded since the last collect."""
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_long()
    check(_lib.lib().fgt_prof_collect_kind(PROF_KINDS[kind], C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "fgt_prof_collect_kind")
    return ms.value, fl.value, n.value, by.value


def mfma_probe(f32=False, iters=20000, device=None):
    """Sustained rate of the matrix cores on random operands and the shader clock they run at under that load (fgt_mfma_probe):
    (TFLOP/s, GHz).  The nominal peaks (2.5 PF bf16, 157.3 TF fp32) assume 2.4 GHz; the chip clocks to its power budget."""
    dev = torch.device(device or "cuda:0")
    if dev.type != "cuda":
        raise RuntimeError("fgt_mfma_probe runs on the MI355X (cuda) device")
    _lib.init_device(dev.index or 0)
    ws = torch.empty(_lib.lib().fgt_mfma_probe_workspace(), dtype=torch.uint8, device=dev)
    tf, ghz = C.c_double(), C.c_double()
    with torch.cuda.device(dev):
        check(_lib.lib().fgt_mfma_probe(int(bool(f32)), int(iters), C.c_void_p(ws.data_ptr()), C.byref(tf), C.byref(ghz), _stream()), "fgt_mfma_probe")
    return tf.value, ghz.value
