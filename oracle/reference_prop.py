"""Runs the REFERENCE'S OWN tool/get_flowNN_gradient.py and tool/utils/Poisson_blend_img.py in this container (TEST
INFRASTRUCTURE ONLY; authoring container only — `available()` is False on the GPU box).

Both files import cv2 at module level; the only cv2 call on the `Nonlocal = False` path of get_flowNN_gradient is
`cv2.remap(..., INTER_LINEAR)` (through tool/utils/common_utils.py:164,250-251), Poisson_blend_img calls none.  A stub `cv2`
module whose `remap` is `oracle.prop_oracle.remap_bilinear` (the documented stand-in, see that file) is installed before the
import, plus `np.bool = bool` (the tool uses the alias numpy >= 1.24 removed: get_flowNN_gradient.py:438).  Everything else
that executes is the reference's source, unmodified.
"""
import importlib
import os
import sys
import types

import numpy as np

REF = os.environ.get("FGT_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "tool", "get_flowNN_gradient.py"))


def _install(tab=32):
    from oracle import prop_oracle
    sys.dont_write_bytecode = True
    if not hasattr(np, "bool"):
        np.bool = bool
    cv2 = sys.modules.get("cv2") or types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.remap = lambda img, mx, my, interpolation=1: prop_oracle.remap_bilinear(img, mx, my, tab)
    if not hasattr(cv2, "setNumThreads"):
        cv2.setNumThreads = lambda n: None
        cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda x: None)
    sys.modules["cv2"] = cv2


_cache = {}


def _import_from_tool(name):
    """Import `name` with tool/ FIRST on sys.path and no foreign `utils` package in sys.modules (RAFT, FGT and LAFC each ship a
    top-level `utils` / `models`; other tests put those trees on sys.path), then restore both."""
    if name in _cache:
        return _cache[name]
    tool = os.path.join(REF, "tool")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "utils" or k.startswith("utils.")}
    sys.path.insert(0, tool)
    try:
        mod = importlib.import_module(name)
    finally:
        sys.path.remove(tool)
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            sys.modules.pop(k)
        sys.modules.update(saved)
    _cache[name] = mod
    return mod


def get_flownn_gradient_fn(tab=32):
    """The reference function `get_flowNN_gradient(args, gradient_x, gradient_y, mask_RGB, mask, videoFlowF, videoFlowB, None, None)`."""
    _install(tab)
    return _import_from_tool("get_flowNN_gradient").get_flowNN_gradient


def poisson_blend_fn():
    """The reference function `Poisson_blend_img(imgTrg, imgSrc_gx, imgSrc_gy, holeMask, gradientMask)` (scipy lsqr inside)."""
    _install()
    return _import_from_tool("utils.Poisson_blend_img").Poisson_blend_img


def run_get_flownn_gradient(gx, gy, mask, flow_f, flow_b, thres=5.0, alpha=0.1, tab=32):
    """Frame-major arrays in (gradients [N,H,W,3], mask [N,H,W] bool, flows [N-1,H,W,2]) -> the reference's outputs, frame-major."""
    import argparse
    import contextlib
    import io
    fn = get_flownn_gradient_fn(tab)
    args = argparse.Namespace(Nonlocal=False, consistencyThres=thres, alpha=alpha)
    to_ref = lambda a: np.ascontiguousarray(np.moveaxis(a, 0, -1))
    with contextlib.redirect_stdout(io.StringIO()):
        ox, oy, fill = fn(args, to_ref(gx).copy(), to_ref(gy).copy(), None, to_ref(mask).astype(bool), to_ref(flow_f), to_ref(flow_b), None, None)
    return np.moveaxis(ox, -1, 0), np.moveaxis(oy, -1, 0), np.moveaxis(fill, -1, 0)
