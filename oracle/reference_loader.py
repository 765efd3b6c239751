"""Import the real reference modules from /root/reference (authoring container only; TEST INFRASTRUCTURE).

Follows SURVEY.md Appendix D: reference on sys.path, read-only tree (no bytecode), a stub `cv2` so that
`import RAFT` works (RAFT/utils/frame_utils.py:6-9 only calls cv2.setNumThreads / cv2.ocl.setUseOpenCL at import).
Nothing on the GPU box may call this: `available()` is False there.
"""
import argparse
import os
import sys
import types

REF = os.environ.get("FGT_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "FGT", "models"))


def _setup():
    sys.dont_write_bytecode = True
    for p in (os.path.join(REF, "LAFC"), os.path.join(REF, "FGT"), REF):
        if p not in sys.path:
            sys.path.insert(0, p) if not p.endswith("LAFC") else sys.path.append(p)
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.setNumThreads = lambda n: None
        cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda x: None)
        sys.modules["cv2"] = cv2


def fgt_model(cfg):
    _setup()
    from FGT.models.model import Model
    return Model(cfg).eval()


def fgt_submodules():
    _setup()
    import FGT.models.model as m
    return m


def lafc_model(cfg):
    _setup()
    from importlib import import_module
    return import_module("LAFC.models.lafc").Model(cfg).eval()


def raft_model(small=False):
    _setup()
    from RAFT import RAFT
    return RAFT(argparse.Namespace(small=small, mixed_precision=False, alternate_corr=False)).eval()


def warp_fns():
    _setup()
    from importlib import import_module
    m = import_module("LAFC.models.utils.fbConsistencyCheck")
    return m.image_warp, m.fbConsistencyCheck
