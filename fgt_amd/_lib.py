"""ctypes binding of libfgt_hip.so (see include/fgt_hip.h).  There is NO fallback: a missing or unloadable
library raises at first use, and every non-zero return code raises RuntimeError with the library's message."""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede loading libfgt_hip.so: torch ships its own libamdhip64; loading ours first would
#                              bring up a second HIP runtime with no device context ("no ROCm-capable device is detected").

_HERE = os.path.dirname(os.path.abspath(__file__))
# FGT_HIP_LIB: diagnostic builds of the same library (fgt_amd.build.build(variant=...): tools/split_sweep.py --diag, tools/conv_trace.py); there is still
# no fallback — a missing file raises.
LIB_PATH = os.environ.get("FGT_HIP_LIB") or os.path.join(_HERE, "lib", "libfgt_hip.so")

ABI_VERSION = 9        # include/fgt_hip.h: what fgt_abi_version() of a matching build returns

ACT = {"none": 0, None: 0, "lrelu": 1, "relu": 2, "sigmoid": 3, "tanh": 4}
EPI = {"none": 0, None: 0, "mul": 1, "add": 2, "gru": 3, "affine": 4, "ps_add2": 5}
PREC = {"fp32": 0, None: 0, "bf16x3": 1, "f16": 2}
TILE = {"auto": 0, None: 0, "128x128": 1, "128x64": 2, "64x64": 3, "128x32": 4, "256x128": 5, "128x128x8": 6, "256x128x16": 7, "256x64x8": 8, "256x128x8s3": 10, "256x128x16s3": 11, "128x128x8s4": 12, "256x128x8pp": 13, "128x128x8pp": 14, "256x128x8il": 16, "256x256p8": 17, "256x128p8": 18, "128x128ea": 26, "128x64ea": 27, "64x64ea": 28, "128x128x8ea": 29, "256x128x16ea": 30, "256x64x8ea": 31, "128x128x8lw": 32, "128x128lw": 33, "128x64lw": 34, "128x128x8xy": 35,
        "256x128ea": 38,      # f16 kernel only
        "c4": 40,             # csrc/conv_c4.hip: bf16x3, fp32 4-channel input, square 3 / 5 / 7 kernel
        # f16 kernel only, FGT_TILE_* + 100: the same tiles on the wide LDS image (128-byte rows, full-line LDS-DMA pieces)
        "128x128w": 101, "128x64w": 102, "64x64w": 103, "128x32w": 104, "256x128w": 105, "128x128x8w": 106, "256x128x16w": 107, "256x64x8w": 108,
        "128x128eaw": 126, "128x64eaw": 127, "64x64eaw": 128, "128x128x8eaw": 129, "256x128x16eaw": 130, "256x64x8eaw": 131, "256x128eaw": 138,
        "256x256p8w": 117, "256x128p8w": 118,     # bf16x3 on interleaved inputs only (csrc/conv_wide.hip)
        "128x128t": 201, "128x64t": 202, "64x64t": 203, "128x64x8t": 204, "128x128x8t": 206, "256x128it": 205, "256x256it": 217, "128x128it": 226,     # FGT_TILE_* + 200: the tap-reusing kernel (csrc/conv_taps.hip)
        "128x128r": 301, "128x64r": 302, "64x64r": 303, "128x128x8r": 306,     # diagnostic builds only: the same with register-fed weights (w_il = 2)
        # only in builds with -DFGT_P8_ABLATIONS (A/B + timing-only instances):
        "256x256p8n": 19, "256x256p8l": 20, "x22nodma": 22, "x23nomma": 23, "x24reads": 24, "x25locknodma": 25}


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "C0", "ld0", "off0", "C1", "ld1", "off1", "Cout", "groups",
                                       "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw", "upsample", "pad_mode",
                                       "in_relu", "Ho", "Wo", "ldo", "ooff", "out_nchw", "act")] + \
               [("slope", C.c_float)] + \
               [(n, C.c_int) for n in ("epi", "act2", "ld_aux1", "ld_aux2")] + \
               [("out_scale", C.c_float)] + \
               [(n, C.c_int) for n in ("Kpad", "Npad", "tile", "precision", "in_split", "out_split", "ldo_s", "ooff_s", "w_il", "k_alg")] + \
               [(n, C.c_longlong) for n in ("ps0", "ps1", "pso")] + \
               [(n, C.c_int) for n in ("ps_r", "ps_c", "ps_g0", "ps_H", "ps_W", "ky_skip_n0", "aux_per_image", "n_alg", "ld_bias", "tile_order", "dual_n0", "ps_phase_pad")] + \
               [(n, C.c_longlong) for n in ("gb_x0", "gb_w", "gb_o")]      # ABI 8


class AttnDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mode", "b", "t", "h", "w", "nh", "nw", "heads", "group", "ws", "n_global",
                                       "ldq", "qoff", "ldk", "koff", "ldv", "voff", "ldg_k", "ldg_v", "ldo", "precision", "out_split")] + \
               [("pso", C.c_longlong)] + [(n, C.c_int) for n in ("in_split", "tq")] + \
               [(n, C.c_longlong) for n in ("psq", "psk", "psv", "psg_k", "psg_v")] + \
               [(n, C.c_int) for n in ("compact", "pad_row")]


_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "fgt_last_error": [],
    "fgt_abi_version": [],
    "fgt_init": [_I],
    "fgt_conv_taps_route": [C.POINTER(ConvDesc)],
    "fgt_conv2d": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "fgt_split": [_P, _L, _I, _I, _P, _I, C.c_longlong, _I, _P],
    "fgt_layernorm": [_P, _I, _I, _P, _I, _I, _L, _F, _P, _P, _P, _I, _P, _P, _P, _I, C.c_longlong, C.c_longlong, _P],
    "fgt_attention": [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P],
    "fgt_dw_pool": [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    "fgt_dw3x3_residual": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "fgt_fold": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, C.c_longlong, _I, _P],
    "fgt_nchw_to_nhwc": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _F, _F, _P],
    "fgt_nhwc_to_nchw": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "fgt_pad_tokens": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "fgt_warp": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "fgt_fb_consistency": [_P, _P, _I, _I, _I, _F, _F, _P, _P, _P],
    "fgt_avgpool2": [_P, _L, _I, _I, _P, _P],
    "fgt_corr_lookup": [C.POINTER(_P), _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "fgt_corr_lookup_split": [C.POINTER(_P), _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, C.c_long, _I, _P],
    "fgt_convex_upsample": [_P, _I, _P, _I, _I, _I, _I, _P, _P],
    "fgt_instnorm_stats": [_P, _I, _I, _I, _I, _P, _P],
    "fgt_instnorm_apply": [_P, _I, _I, _I, _I, _P, _F, _I, _P, _I, _I, _P, _I, _P],
    "fgt_instnorm_apply_split": [_P, _I, _I, _I, _I, _P, _F, _I, _P, _I, _I, _P, _I, _P, _I, C.c_longlong, _P],
    "fgt_axpby": [_P, _I, _F, _P, _I, _F, _L, _I, _I, _F, _P, _I, _P],
    "fgt_compose_blend": [_P, _P, _P, _I, _P, _P, _I, _I, _P, _P],
    "fgt_compose_blend_u8": [_P, _P, _P, _I, _P, _P, _I, _I, _P, _P],
    "fgt_quantize_u8": [_P, _L, _P, _P],
    "fgt_pack_frames": [_P, _P, _P, _I, _I, _I, _P, _I, _P],
    "fgt_norm_flows": [_P, _I, _I, _I, _L, _P, _P],
    "fgt_gather_rows": [_P, _L, _P, _I, _L, _P, _L, _P],
    "fgt_laplace_fill_workspace": [_I, _I, _I],
    "fgt_laplace_fill": [_P, _P, _I, _I, _I, _I, _P, _P, _I, _F, _P],
    "fgt_flow_propagate_workspace": [_I, _I, _I],
    "fgt_flow_propagate": [_P, _P, _P, _P, _P, _I, _I, _I, C.c_double, C.c_double, _I, _P, _P, _P, _P, _P],
    "fgt_poisson_blend_workspace": [_I, _I, _I],
    "fgt_poisson_blend": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "fgt_mask_bbox": [_P, _I, _I, _I, _P, _P],
    "fgt_laplace_fill_onchip": [_P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _F, _P, _P],
    "fgt_poisson_blend_onchip": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "fgt_mfma_probe_workspace": [],
    "fgt_mfma_probe": [_I, _I, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), _P],
    "fgt_prof_enable": [_I],
    "fgt_prof_enable_kinds": [C.c_uint],
    "fgt_prof_collect": [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)],
    "fgt_prof_collect_kind": [_I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)],
}
_RESTYPES = {"fgt_last_error": C.c_char_p, "fgt_prof_enable": None, "fgt_prof_enable_kinds": None, "fgt_laplace_fill_workspace": C.c_long, "fgt_flow_propagate_workspace": C.c_long, "fgt_poisson_blend_workspace": C.c_long, "fgt_mfma_probe_workspace": C.c_long}

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -m fgt_amd.build). "
                "fgt_amd has no CPU/PyTorch fallback.")
        h = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the .so does not export a declared symbol
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = h
    return _lib


_inited = set()


def init_device(index):
    """fgt_init(device) once per device: allocates the library's zero page outside any stream capture."""
    if index not in _inited:
        check(lib().fgt_init(int(index)), "fgt_init")
        _inited.add(index)


def check(rc, what):
    if rc != 0:
        msg = lib().fgt_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
