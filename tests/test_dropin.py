"""The drop-in boundary: with fgt_amd/dropin on sys.path the reference tool's imports
(tool/video_inpainting.py:4-6,17,201-230) resolve to the MI355X modules, and the C struct layouts that the
ctypes binding assumes are the ones the C compiler produces for include/fgt_hip.h."""
import ctypes
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_paths_resolve_to_fgt_amd():
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from importlib import import_module\n"
            "M = import_module('FGT.models.' + 'model').Model\n"
            "L = import_module('LAFC.models.' + 'lafc').Model\n"
            "from RAFT import RAFT\n"
            "print(M.__module__, L.__module__, RAFT.__module__)\n") % os.path.join(ROOT, "fgt_amd", "dropin")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=tempfile.gettempdir())
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["fgt_amd.fgt_model", "fgt_amd.lafc_model", "fgt_amd.raft_model"]


def test_ctypes_struct_layout_matches_the_c_header():
    from fgt_amd import _lib
    fields = {"fgt_conv_desc": _lib.ConvDesc, "fgt_attn_desc": _lib.AttnDesc}
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % os.path.join(ROOT, "include", "fgt_hip.h"), "int main(){"]
    for cname, st in fields.items():
        for fname, _ in st._fields_:
            prog.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
        prog.append('printf("%s.sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
    prog.append("return 0;}")
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "l.c"), os.path.join(td, "l")
        open(src, "w").write("\n".join(prog))
        subprocess.run(["gcc", src, "-o", exe], check=True)
        out = dict(line.split() for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, st in fields.items():
        for fname, _ in st._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"
        assert int(out[f"{cname}.sizeof"]) == ctypes.sizeof(st)
